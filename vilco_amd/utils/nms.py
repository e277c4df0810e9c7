"""1-D NMS on the MI355X, behind the reference's two NMS surfaces:

 * `nms_1d_cpu` -- an object with the compiled extension's functions
   `nms(segs, scores, iou_threshold)` / `softnms(segs, scores, dets, iou_threshold, sigma,
   min_score, method)` (MQ/libs/utils/csrc/nms_cpu.cpp:172-182), same return values, same
   error text for non-contiguous / non-float inputs; the work runs in libvilco_hip.so.
 * `batched_nms(...)` (MQ/libs/utils/nms.py:103-190): the per-class python loop becomes ONE
   kernel launch (one workgroup per class) and nothing leaves the device until the caller asks.
"""
import torch

from .. import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_input(t, name):
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float but found %s" % str(t.dtype).replace("torch.", "").capitalize())


def _dev(t):
    return t if t.is_cuda else t.cuda()


def _run_hard(segs, scores, seg_off, nseg, iou_threshold):
    lib = _lib.load()
    n = segs.shape[0]
    out_idx = torch.empty(max(n, 1), dtype=torch.int64, device=segs.device)
    out_cnt = torch.zeros(max(nseg, 1), dtype=torch.int64, device=segs.device)
    wsz = lib.vilco_nms_workspace(n, nseg)
    ws = torch.empty(wsz, dtype=torch.uint8, device=segs.device)
    _lib.check(lib.vilco_nms_1d(segs.data_ptr(), scores.data_ptr(), seg_off.data_ptr(), nseg, n,
                                float(iou_threshold), out_idx.data_ptr(), out_cnt.data_ptr(),
                                ws.data_ptr(), wsz, _stream()))
    return out_idx, out_cnt


def _run_soft(segs, scores, seg_off, nseg, iou_threshold, sigma, min_score, method, max_num):
    lib = _lib.load()
    n = segs.shape[0]
    dets = torch.empty(max(n, 1), 3, dtype=torch.float32, device=segs.device)
    out_idx = torch.empty(max(n, 1), dtype=torch.int64, device=segs.device)
    out_cnt = torch.zeros(max(nseg, 1), dtype=torch.int64, device=segs.device)
    wsz = lib.vilco_nms_workspace(n, nseg)
    ws = torch.empty(wsz, dtype=torch.uint8, device=segs.device)
    _lib.check(lib.vilco_softnms_1d(segs.data_ptr(), scores.data_ptr(), seg_off.data_ptr(), nseg, n,
                                    float(iou_threshold), float(sigma), float(min_score), int(method),
                                    int(max_num), dets.data_ptr(), out_idx.data_ptr(), out_cnt.data_ptr(),
                                    ws.data_ptr(), wsz, _stream()))
    return dets, out_idx, out_cnt


SOFT_KERNELS = {"auto": -1, "reg": 0, "rows": 1, "legacy": 2}


class soft_kernel:
    """`with soft_kernel("rows"): ...` -- no soft-NMS device kernel faster than the named one inside the block
    (vilco_nms_set_kernel; include/vilco_hip.h).  Every kernel returns the reference's indices bit for bit; the switch
    exists so that tests and measurements can run each of them on the same inputs."""

    def __init__(self, kind):
        self.kind = SOFT_KERNELS[kind] if isinstance(kind, str) else int(kind)

    def __enter__(self):
        self.prev = _lib.load().vilco_nms_set_kernel(self.kind)
        return self

    def __exit__(self, *exc):
        _lib.load().vilco_nms_set_kernel(self.prev)
        return False


def last_soft_kernels():
    """names of the device kernels the last soft-NMS call launched"""
    m = _lib.load().vilco_nms_last_kernels()
    return {k for k, v in SOFT_KERNELS.items() if v >= 0 and (m >> v) & 1}


class _Nms1dModule:
    """stand-in for the compiled python module `nms_1d_cpu`"""

    @staticmethod
    def nms(segs, scores, iou_threshold):
        _check_input(segs, "segs")
        _check_input(scores, "scores")
        if segs.numel() == 0:
            return torch.empty(0, dtype=torch.int64, device=segs.device)
        d_segs, d_scores = _dev(segs), _dev(scores)
        off = torch.tensor([0, segs.shape[0]], dtype=torch.int64, device=d_segs.device)
        idx, cnt = _run_hard(d_segs, d_scores, off, 1, iou_threshold)
        return idx[: int(cnt[0].item())].to(segs.device)

    @staticmethod
    def softnms(segs, scores, dets, iou_threshold, sigma, min_score, method):
        _check_input(segs, "segs")
        _check_input(scores, "scores")
        _check_input(dets, "dets")
        if segs.numel() == 0:
            return torch.empty(0, dtype=torch.int64, device=segs.device)
        d_segs, d_scores = _dev(segs), _dev(scores)
        off = torch.tensor([0, segs.shape[0]], dtype=torch.int64, device=d_segs.device)
        d, idx, cnt = _run_soft(d_segs, d_scores, off, 1, iou_threshold, sigma, min_score, method, 0)
        k = int(cnt[0].item())
        dets[:k].copy_(d[:k])        # rows 0..K-1 are written in place, like the extension
        return idx[:k].to(segs.device)


nms_1d_cpu = _Nms1dModule()


def seg_voting(nms_segs, all_segs, all_scores, iou_threshold, score_offset=1.5):
    """segment voting (nms.py:67-101); class-agnostic path only, tiny -> device tensor ops."""
    left = torch.maximum(nms_segs[:, None, 0], all_segs[None, :, 0])
    right = torch.minimum(nms_segs[:, None, 1], all_segs[None, :, 1])
    inter = (right - left).clamp(min=0)
    iou = inter / ((nms_segs[:, None, 1] - nms_segs[:, None, 0]) + (all_segs[None, :, 1] - all_segs[None, :, 0]) - inter)
    w = (iou >= iou_threshold).to(all_scores.dtype) * all_scores[None, :] * iou
    w = w / torch.sum(w, dim=1, keepdim=True)
    return w @ all_segs


@torch.no_grad()
def batched_nms(segs, scores, cls_idxs, iou_threshold, min_score, max_seg_num, use_soft_nms=True,
                multiclass=True, sigma=0.5, voting_thresh=0.75):
    """same contract as the reference; inputs may live on the host or the device, outputs follow the
    inputs' device."""
    in_dev = segs.device
    if segs.shape[0] == 0:
        return (torch.zeros([0, 2], device=in_dev), torch.zeros([0, ], device=in_dev),
                torch.zeros([0, ], dtype=cls_idxs.dtype, device=in_dev))
    segs, scores, cls_idxs = _dev(segs).contiguous().float(), _dev(scores).contiguous().float(), _dev(cls_idxs)
    all_segs, all_scores = segs, scores

    if not use_soft_nms and min_score > 0:       # NMSop filters by score first (nms.py:15-22)
        keep = scores > min_score
        segs, scores, cls_idxs = segs[keep].contiguous(), scores[keep].contiguous(), cls_idxs[keep]

    if multiclass:
        # stable sort by class == the reference's `for class_id in torch.unique(cls_idxs)` +
        # `torch.where(cls_idxs == class_id)` (original order inside each class)
        order = torch.sort(cls_idxs, stable=True).indices
        segs, scores, cls_sorted = segs[order].contiguous(), scores[order].contiguous(), cls_idxs[order]
        _, counts = torch.unique_consecutive(cls_sorted, return_counts=True)
        seg_off = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=segs.device)
        seg_off[1:] = torch.cumsum(counts, 0)
    else:
        cls_sorted = cls_idxs
        seg_off = torch.tensor([0, segs.shape[0]], dtype=torch.int64, device=segs.device)
    nseg = seg_off.numel() - 1
    n = segs.shape[0]
    if n == 0:
        return (torch.zeros([0, 2], device=in_dev), torch.zeros([0, ], device=in_dev),
                torch.zeros([0, ], dtype=cls_idxs.dtype, device=in_dev))

    if use_soft_nms:
        dets, idx, cnt = _run_soft(segs, scores, seg_off, nseg, iou_threshold, sigma, min_score, 2,
                                   max_seg_num if max_seg_num > 0 else 0)
    else:
        dets = None
        idx, cnt = _run_hard(segs, scores, seg_off, nseg, iou_threshold)
    if max_seg_num > 0:
        cnt = torch.clamp(cnt[:nseg], max=max_seg_num)
    else:
        cnt = cnt[:nseg]
    # flat positions of the kept rows: class c contributes seg_off[c] + [0, cnt[c])
    pos = torch.arange(n, device=segs.device)
    cls_of_pos = torch.searchsorted(seg_off[1:], pos, right=True)
    local = pos - seg_off[cls_of_pos]
    sel = local < cnt[cls_of_pos]
    src = seg_off[cls_of_pos][sel] + idx[:n][sel]       # global (class-sorted) index of each pick
    if use_soft_nms:
        new_segs, new_scores = dets[:n][sel][:, :2], dets[:n][sel][:, 2]
    else:
        new_segs, new_scores = segs[src], scores[src]
    new_cls = cls_sorted[src]

    if (not multiclass) and voting_thresh > 0:
        new_segs = seg_voting(new_segs, all_segs, all_scores, voting_thresh)

    order = torch.sort(new_scores, descending=True, stable=True).indices
    k = min(max_seg_num, new_segs.shape[0])
    order = order[:k]
    return (new_segs[order].clone().to(in_dev), new_scores[order].clone().to(in_dev),
            new_cls[order].clone().to(in_dev))
