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
    if ref is not None and not ties:      # tie order of aten's unstable CPU sort is implementation-defined
        want = ref.nms(torch.from_numpy(segs), torch.from_numpy(scores), 0.5).numpy()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [1, 2, 8, 257, 1200, 5000])
@pytest.mark.parametrize("sigma,min_score", [(0.5, 0.001), (0.75, 0.01), (0.99, 0.2)])
@pytest.mark.parametrize("ties", [False, True])
def test_soft_nms(dev, n, sigma, min_score, ties):
    from oracle import nms_oracle
    from vilco_amd.utils.nms import nms_1d_cpu
    segs, scores = make_case(n, 7 * n + 3, ties)
    dets = torch.zeros(n, 3)
    got = nms_1d_cpu.softnms(torch.from_numpy(segs), torch.from_numpy(scores), dets, 0.1, sigma, min_score, 2).numpy()
    if n <= 1200:
        want, wdets = nms_oracle.softnms(segs, scores, 0.1, sigma, min_score, 2)
        assert np.array_equal(got, want)
        np.testing.assert_allclose(dets.numpy()[:len(got)], wdets, rtol=1e-5, atol=1e-7)
    ref = ref_module()
    if ref is not None:
        rdets = torch.zeros(n, 3)
        want = ref.softnms(torch.from_numpy(segs), torch.from_numpy(scores), rdets, 0.1, sigma, min_score, 2).numpy()
        assert np.array_equal(got, want)
        np.testing.assert_allclose(dets.numpy()[:len(got)], rdets.numpy()[:len(got)], rtol=1e-5, atol=1e-7)


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
