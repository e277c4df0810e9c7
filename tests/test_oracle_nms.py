"""CPU: pin the numpy NMS restatement against the reference extension (oracle/_ref, compiled from
the reference's own nms_cpu.cpp) when it is available, and against the committed goldens."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import build_ref, nms_oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def make_case(n, seed, ties=False):
    g = np.random.RandomState(seed)
    c = g.uniform(0, 200.0, n).astype(np.float32)
    w = g.uniform(0.5, 30, n).astype(np.float32)
    segs = np.stack([c - w / 2, c + w / 2], 1).astype(np.float32)
    scores = g.uniform(0.001, 1, n).astype(np.float32)
    if ties and n > 4:
        scores[n // 2:] = scores[: n - n // 2]
        segs[-1] = segs[0]
    return segs, scores


@pytest.mark.parametrize("n", [0, 1, 8, 257, 600])
@pytest.mark.parametrize("ties", [False, True])
def test_against_reference_build(n, ties):
    ref = build_ref.load_ref()
    if ref is None:
        pytest.skip("reference extension not built (no /root/reference and no prebuilt oracle/_ref)")
    segs, scores = make_case(n, 11 + n, ties)
    ts, tc = torch.from_numpy(segs).reshape(-1, 2), torch.from_numpy(scores)
    if not ties:
        # exact score ties: aten's CPU sort (not requested stable, nms_cpu.cpp:28) orders them by
        # std::sort internals for n > 16 -- implementation-defined, so hard-NMS parity is tie-free only
        assert np.array_equal(nms_oracle.nms(segs, scores, 0.4), ref.nms(ts, tc, 0.4).numpy())
    for sigma, ms in ((0.5, 0.001), (0.99, 0.2)):
        dets = torch.zeros(n, 3)
        want = ref.softnms(ts, tc, dets, 0.1, sigma, ms, 2).numpy()
        got, gdets = nms_oracle.softnms(segs, scores, 0.1, sigma, ms, 2)
        assert np.array_equal(got, want)
        np.testing.assert_allclose(gdets, dets.numpy()[:len(want)], rtol=2e-6, atol=1e-8)


def test_against_goldens():
    files = sorted(glob.glob(os.path.join(HERE, "golden", "nms_*.npz")))
    assert files, "golden NMS fixtures missing"
    for f in files:
        z = np.load(f)
        if z["kind"] == "hard":
            assert np.array_equal(nms_oracle.nms(z["segs"], z["scores"], float(z["thr"])), z["inds"]), f
        elif z["kind"] == "soft":
            got, dets = nms_oracle.softnms(z["segs"], z["scores"], float(z["thr"]), float(z["sigma"]),
                                           float(z["min_score"]), 2)
            assert np.array_equal(got, z["inds"]), f
            np.testing.assert_allclose(dets, z["dets"][:len(got)], rtol=2e-6, atol=1e-8)
        elif z["kind"] == "voting":
            # class-agnostic + segment voting (nms.py:161-180): recorded by tests/golden/make_golden_nms_voting.py
            s, sc, c = nms_oracle.batched_nms(z["segs"], z["scores"], z["cls"], float(z["thr"]), float(z["min_score"]),
                                              int(z["max_seg_num"]), bool(z["soft"]), False, float(z["sigma"]),
                                              float(z["voting_thresh"]))
            assert np.array_equal(c, z["out_cls"]), f
            np.testing.assert_allclose(s, z["out_segs"], rtol=2e-5)          # (a weighted mean over ~100 neighbours: summation order)
            np.testing.assert_allclose(sc, z["out_scores"], rtol=1e-5, atol=1e-8)
        else:
            s, sc, c = nms_oracle.batched_nms(z["segs"], z["scores"], z["cls"], float(z["thr"]), float(z["min_score"]),
                                              int(z["max_seg_num"]), bool(z["soft"]), True, float(z["sigma"]), 0.0)
            assert np.array_equal(c, z["out_cls"]), f
            np.testing.assert_allclose(s, z["out_segs"], rtol=1e-6)
            np.testing.assert_allclose(sc, z["out_scores"], rtol=1e-5, atol=1e-8)


def test_expf_matches_libm():
    """the restated glibc expf (oracle + HIP kernel use the same algorithm) is bit-identical to libm"""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype, libm.expf.argtypes = ctypes.c_float, [ctypes.c_float]
    rng = np.random.RandomState(0)
    xs = np.concatenate([-rng.uniform(0, 2.5, 40000), -rng.uniform(0, 1e-3, 5000), -rng.uniform(0, 70, 5000)]).astype(np.float32)
    want = np.array([libm.expf(float(x)) for x in xs], dtype=np.float32)
    assert np.array_equal(nms_oracle.expf_libm(xs), want)
