"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's compiled 1-D NMS.

Follows MQ/libs/utils/csrc/nms_cpu.cpp: `nms` = nms_1d_cpu (:19-58), `softnms` = softnms_1d_cpu
(:67-160), and the python wrapper MQ/libs/utils/nms.py: `batched_nms` (:103-190) with NMSop (:8-35) /
SoftNMSop (:38-64).  float32 arithmetic is kept step by step (np.float32 scalars) so scores and
tie-breaks round like the C++.  Parity pinned: tests/test_oracle_nms.py checks these functions against
the reference extension itself (oracle/_ref/nms_1d_cpu.so, built from the reference source) and
against the committed goldens tests/golden/nms_*.npz generated from it.
"""
import numpy as np

f32 = np.float32

# std::exp(float) of the reference == glibc expf (2.35 in this image): the table-driven fp64 algorithm of the ARM
# optimized routines (32-entry 2^(i/32) table, cubic polynomial, one rounding to float).  numpy's own SIMD exp
# can differ from it by 1 ulp, which is enough to flip a soft-NMS pick among 30 000 candidates, so the algorithm
# is restated here (test_expf_matches_libm pins it against libm itself).
_N = 32
_TAB = None


def _tab():
    global _TAB
    if _TAB is None:
        from decimal import Decimal, getcontext
        import struct
        getcontext().prec = 80
        t = []
        for i in range(_N):
            v = float(Decimal(2) ** (Decimal(i) / Decimal(_N)))
            t.append((struct.unpack('<Q', struct.pack('<d', v))[0] - (i << 47)) & 0xFFFFFFFFFFFFFFFF)
        _TAB = np.array(t, dtype=np.uint64)
    return _TAB


def expf_libm(x):
    """float32 -> float32, bit-identical to glibc expf for |x| < 80 (vectorised)."""
    xd = np.asarray(x, dtype=np.float32).astype(np.float64)
    z = (float.fromhex('0x1.71547652b82fep+0') * _N) * xd
    shift = float.fromhex('0x1.8p+52')
    kd = z + shift
    ki = kd.view(np.uint64) if isinstance(kd, np.ndarray) else np.float64(kd).view(np.uint64)
    kd = kd - shift
    r = z - kd
    t = _tab()[(ki % np.uint64(_N)).astype(np.int64)] + (ki << np.uint64(47))
    s = t.view(np.float64) if isinstance(t, np.ndarray) else np.uint64(t).view(np.float64)
    c0 = float.fromhex('0x1.c6af84b912394p-5') / _N / _N / _N
    c1 = float.fromhex('0x1.ebfce50fac4f3p-3') / _N / _N
    c2 = float.fromhex('0x1.62e42ff0c52d6p-1') / _N
    y = (c0 * r + c1) * (r * r) + (c2 * r + 1.0)
    return (y * s).astype(np.float32)


def nms(segs, scores, iou_threshold):
    segs = np.asarray(segs, dtype=f32).reshape(-1, 2)
    scores = np.asarray(scores, dtype=f32)
    n = segs.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    x1, x2 = segs[:, 0].copy(), segs[:, 1].copy()
    areas = (x2 - x1 + f32(1e-6)).astype(f32)
    order = np.argsort(-scores, kind="stable")          # aten CPU sort(descending=True) is stable
    keep = np.ones(n, dtype=bool)
    thr = f32(iou_threshold)
    for _i in range(n):
        if not keep[_i]:
            continue
        i = order[_i]
        for _j in range(_i + 1, n):
            if not keep[_j]:
                continue
            j = order[_j]
            inter = max(f32(0), f32(min(x2[i], x2[j]) - max(x1[i], x1[j])))
            ovr = f32(inter / f32(f32(areas[i] + areas[j]) - inter))
            if ovr >= thr:
                keep[_j] = False
    return order[keep].astype(np.int64)


def softnms(segs, scores, iou_threshold, sigma, min_score, method, max_num=0):
    """returns (inds, dets[K,3]).  max_num > 0 stops after max_num picks (exact for the caller)."""
    segs = np.asarray(segs, dtype=f32).reshape(-1, 2)
    n = segs.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64), np.zeros((0, 3), dtype=f32)
    x1, x2 = segs[:, 0].copy(), segs[:, 1].copy()
    sc = np.asarray(scores, dtype=f32).copy()
    areas = (x2 - x1 + f32(1e-6)).astype(f32)
    inds = np.arange(n, dtype=np.int64)
    dets = np.zeros((n, 3), dtype=f32)
    thr, sigma, min_score = f32(iou_threshold), f32(sigma), f32(min_score)
    nsegs, i = n, 0
    while i < nsegs:
        if max_num > 0 and i >= max_num:
            break
        mp = i + int(np.argmax(sc[i:nsegs]))             # first maximum == strict '<' scan
        for arr in (x1, x2, sc, areas, inds):
            arr[i], arr[mp] = arr[mp], arr[i]
        dets[i] = (x1[i], x2[i], sc[i])
        ix1, ix2, ia = x1[i], x2[i], areas[i]
        pos = i + 1
        while pos < nsegs:
            inter = max(f32(0), f32(min(ix2, x2[pos]) - max(ix1, x1[pos])))
            ovr = f32(inter / f32(f32(ia + areas[pos]) - inter))
            w = f32(1)
            if method == 0:
                if ovr >= thr:
                    w = f32(0)
            elif method == 1:
                if ovr >= thr:
                    w = f32(1) - ovr
            elif method == 2:
                w = f32(expf_libm(np.array([f32(-(ovr * ovr) / sigma)], dtype=f32))[0])
            sc[pos] = f32(sc[pos] * w)
            if sc[pos] < min_score:
                last = nsegs - 1
                for arr in (x1, x2, sc, areas, inds):
                    arr[pos] = arr[last]
                nsegs -= 1
                pos -= 1
            pos += 1
        i += 1
    return inds[:i].copy(), dets[:i].copy()


def seg_voting(nms_segs, all_segs, all_scores, iou_threshold):
    """segment voting of the class-agnostic branch (MQ/libs/utils/nms.py:67-101): every kept segment becomes the
    score x IoU weighted mean of ALL input segments whose IoU with it reaches the threshold (the reference's `score_offset`
    is computed and never used, :75).  fp32 throughout, like the reference's torch ops."""
    a = np.asarray(nms_segs, dtype=f32).reshape(-1, 2)
    b = np.asarray(all_segs, dtype=f32).reshape(-1, 2)
    sc = np.asarray(all_scores, dtype=f32)
    left = np.maximum(a[:, None, 0], b[None, :, 0])
    right = np.minimum(a[:, None, 1], b[None, :, 1])
    inter = np.maximum(right - left, f32(0))
    iou = inter / ((a[:, None, 1] - a[:, None, 0]) + (b[None, :, 1] - b[None, :, 0]) - inter)
    w = (iou >= f32(iou_threshold)).astype(f32) * sc[None, :] * iou
    w = w / np.sum(w, axis=1, keepdims=True, dtype=f32)
    return (w @ b).astype(f32)


def batched_nms(segs, scores, cls_idxs, iou_threshold, min_score, max_seg_num, use_soft_nms=True,
                multiclass=True, sigma=0.5, voting_thresh=0.75):
    segs = np.asarray(segs, dtype=f32).reshape(-1, 2)
    scores = np.asarray(scores, dtype=f32)
    cls_idxs = np.asarray(cls_idxs)
    if segs.shape[0] == 0:
        return np.zeros((0, 2), f32), np.zeros((0,), f32), np.zeros((0,), cls_idxs.dtype)

    def one(s, sc, c):
        if use_soft_nms:
            inds, dets = softnms(s, sc, iou_threshold, sigma, min_score, 2)
            k = min(len(inds), max_seg_num) if max_seg_num > 0 else len(inds)
            return dets[:k, :2], dets[:k, 2], c[inds][:k]
        if min_score > 0:
            m = sc > f32(min_score)
            s, sc, c = s[m], sc[m], c[m]
        inds = nms(s, sc, iou_threshold)
        if max_seg_num > 0:
            inds = inds[:min(max_seg_num, len(inds))]
        return s[inds], sc[inds], c[inds]

    if multiclass:
        parts = [one(segs[cls_idxs == k], scores[cls_idxs == k], cls_idxs[cls_idxs == k]) for k in np.unique(cls_idxs)]
        new_segs = np.concatenate([p[0] for p in parts])
        new_scores = np.concatenate([p[1] for p in parts])
        new_cls = np.concatenate([p[2] for p in parts])
    else:
        new_segs, new_scores, new_cls = one(segs, scores, cls_idxs)
        if voting_thresh > 0:
            new_segs = seg_voting(new_segs, segs, scores, voting_thresh)
    order = np.argsort(-new_scores, kind="stable")[:min(max_seg_num, new_segs.shape[0])]
    return new_segs[order], new_scores[order], new_cls[order]
