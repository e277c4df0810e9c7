"""ops.qkv_pre (vilco_qkv_pre_fwd / _bwd): LN1 + the three depthwise k=3 convs + their LayerNorms of MaskedMHCA
(blocks.py:561-563, 363-369) in one launch, against a float64 PyTorch restatement of those reference lines."""
import pytest
import torch
import torch.nn.functional as F

from parity_util import rel_err

pytestmark = pytest.mark.gpu


def _ref(x, g1, b1, ws, gs, bs, lens, stride, eps=1e-5):
    """channel-first reference arithmetic on token-major tensors: x [B,T,C]"""
    B, T, C = x.shape

    def ln(t, g, b):
        r = t - t.mean(dim=2, keepdim=True)
        return r / torch.sqrt((r ** 2).mean(dim=2, keepdim=True) + eps) * g.view(1, 1, C) + b.view(1, 1, C)
    h = ln(x, g1, b1)
    mask = (stride * torch.arange(T // stride)[None, :] < lens[:, None]).to(x.dtype)[:, :, None]       # mask[::stride]
    outs = []
    for w, g, b in zip(ws, gs, bs):
        c = F.conv1d(h.transpose(1, 2), w, None, stride=stride, padding=1, groups=C).transpose(1, 2) * mask
        outs.append(ln(c, g, b))
    return outs, h


@pytest.mark.parametrize("C,stride,want_h", [(256, 1, True), (512, 2, False), (1024, 1, False), (1024, 2, True),
                                             (2304, 1, True), (2304, 2, False), (1536, 1, True), (2048, 2, True), (1280, 1, False)])
def test_qkv_pre_fwd_bwd(dev, C, stride, want_h):
    _check(dev, 2, 48, C, stride, want_h)


@pytest.mark.parametrize("C", [256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2304])
@pytest.mark.parametrize("stride,want_h", [(1, False), (2, True)])
def test_qkv_pre_ring_kernel_all_widths(dev, C, stride, want_h):
    """2 x 320 tokens: enough rows for the ring kernel (producer / consumer waves around an LDS row ring; one instantiation
    per C / 256 and stride), with several runs per clip, a ragged clip and the zero-padding rows at both clip ends"""
    _check(dev, 2, 320, C, stride, want_h)


@pytest.mark.parametrize("B,T,C,stride", [(2, 322, 512, 2), (3, 100, 256, 1), (5, 97, 768, 1), (1, 514, 1024, 2), (2, 131, 2304, 1)])
def test_qkv_pre_ring_kernel_odd_shapes(dev, B, T, C, stride):
    """ring kernel at awkward sizes: runs that do not divide T, odd T at stride 1, a
    single clip, more clips than pipeline depth"""
    _check(dev, B, T, C, stride, True, lens=[T - 7 * (b % 3) for b in range(B)])


@pytest.mark.parametrize("stride,want_h", [(1, True), (2, False)])
def test_qkv_pre_at_the_target_shape(dev, stride, want_h):
    """the shape the north-star HBM target is quoted on (and bench.py times): [2, T = 2304, C = 2304], forward and
    backward against the float64 restatement"""
    _check(dev, 2, 2304, 2304, stride, want_h)


def _check(dev, B, T, C, stride, want_h, lens=None):
    from vilco_amd import ops
    torch.manual_seed(C + stride)
    x = torch.randn(B, T, C, dtype=torch.float64)
    lens = torch.tensor([T, T - 13] if lens is None else lens)
    for b in range(B):
        x[b, int(lens[b]):] = 0.0                         # inputs are masked upstream
    g1, b1 = 1 + 0.1 * torch.randn(C, dtype=torch.float64), 0.1 * torch.randn(C, dtype=torch.float64)
    ws = [0.5 * torch.randn(C, 1, 3, dtype=torch.float64) for _ in range(3)]
    gs = [1 + 0.1 * torch.randn(C, dtype=torch.float64) for _ in range(3)]
    bs = [0.1 * torch.randn(C, dtype=torch.float64) for _ in range(3)]
    leaves = [x, g1, b1] + ws + gs + bs
    for t in leaves:
        t.requires_grad_(True)
    want, want_hh = _ref(x, g1, b1, ws, gs, bs, lens, stride)
    wts = [torch.randn_like(o) for o in want]
    wh = torch.randn_like(want_hh)
    loss = sum((o * w).sum() for o, w in zip(want, wts))
    if want_h:
        loss = loss + (want_hh * wh).sum()
    loss.backward()

    def dv(t, shape=None):
        t = t.detach().float().to(dev)
        return (t.reshape(shape) if shape else t).contiguous().requires_grad_(True)
    xg = dv(x)
    g1g, b1g = dv(g1, (1, C, 1)), dv(b1, (1, C, 1))                      # the reference's [1,C,1] LayerNorm parameters
    wsg = [dv(w) for w in ws]
    gsg, bsg = [dv(g, (1, C, 1)) for g in gs], [dv(b, (1, C, 1)) for b in bs]
    outs = ops.qkv_pre(xg, (g1g, b1g, 1e-5), tuple(wsg), ((gsg[0], bsg[0]), (gsg[1], bsg[1]), (gsg[2], bsg[2]), 1e-5),
                       lens.to(torch.int32).to(dev), stride, want_h)
    assert len(outs) == (4 if want_h else 3)
    loss = sum((o * w.float().to(dev)).sum() for o, w in zip(outs[:3], wts))
    if want_h:
        loss = loss + (outs[3] * wh.float().to(dev)).sum()
    loss.backward()
    for j in range(3):
        assert rel_err(outs[j], want[j]) < 2e-5, ("y", j)
    if want_h:
        assert rel_err(outs[3], want_hh) < 2e-5
    assert rel_err(xg.grad, x.grad) < 1e-4
    assert rel_err(g1g.grad.reshape(-1), g1.grad) < 1e-4 and rel_err(b1g.grad.reshape(-1), b1.grad) < 1e-4
    for j in range(3):
        assert rel_err(wsg[j].grad, ws[j].grad) < 1e-4, ("dw", j)
        assert rel_err(gsg[j].grad.reshape(-1), gs[j].grad) < 1e-4, ("dgamma", j)
        assert rel_err(bsg[j].grad.reshape(-1), bs[j].grad) < 1e-4, ("dbeta", j)


def test_block_fused_equals_unfused(dev):
    """a TransformerBlock at a fusable width: ops.qkv_pre path vs the separate LayerNorm / dwconv kernels"""
    import vilco_amd.modeling as vm
    from vilco_amd import ops
    torch.manual_seed(4)
    C, H, B, T = 256, 4, 2, 64
    for stride in (1, 2):
        blk = vm.TransformerBlock(C, H, n_ds_strides=(stride, stride), path_pdrop=0.1, use_cross_modal=False).to(dev).eval()
        x = torch.randn(B, T, C, device=dev)
        lens = torch.tensor([T, T - 9], dtype=torch.int32, device=dev)
        ops.mask_rows_(x, lens)
        res = []
        for fused in (True, False):
            ops.use_qkv_pre = fused
            try:
                xx = x.clone().requires_grad_(True)
                blk.zero_grad(set_to_none=True)
                y, _ = blk.forward_tm(xx, lens)
                (y * torch.sin(torch.arange(y.numel(), device=dev).view_as(y) * 0.01)).sum().backward()
                res.append((y.detach(), xx.grad, {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None}))
            finally:
                ops.use_qkv_pre = True
        (ya, ga, pa), (yb, gb, pb) = res
        assert rel_err(ya, yb) < 1e-5 and rel_err(ga, gb) < 1e-4
        assert sorted(pa) == sorted(pb)
        for k in pa:
            if not k.endswith(('key_norm.bias', '.key.bias')):
                assert rel_err(pa[k], pb[k], 1e-7) < 1e-4, k
