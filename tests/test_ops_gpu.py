"""HIP kernels (through the C ABI) vs plain fp64 PyTorch references of the same op, fwd + bwd.
Tolerances (max abs error / max abs reference): GEMM-backed ops 2e-5 in the default fp16x2 mode and in the
three-part bf16 split mode (both fp32-equivalent), 2e-4 in the two-part bf16 split mode, 2e-2 in plain bf16;
elementwise 2e-5.  The module runs under the default precision; VILCO_PRECISION=split3 reruns it in that mode."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_GEMM = 2e-5
TOL_EW = 2e-5


def rel(got, want):
    want = want.double().cpu()
    got = got.double().cpu()
    return ((got - want).abs().max() / (want.abs().max() + 1e-12)).item()


def run_pair(fn_hip, fn_ref, inputs, dev, tol, grad_inputs=None, seed=0):
    """inputs: dict name -> cpu tensor (float64 reference copies are made); returns nothing, asserts."""
    g = torch.Generator().manual_seed(seed)
    cpu = {k: (v.double().requires_grad_(v.is_floating_point() and (grad_inputs is None or k in grad_inputs)))
           for k, v in inputs.items()}
    gpu = {k: (v.to(dev).requires_grad_(v.is_floating_point() and (grad_inputs is None or k in grad_inputs)))
           for k, v in inputs.items()}
    want = fn_ref(**cpu)
    got = fn_hip(**gpu)
    assert got.shape == want.shape, (got.shape, want.shape)
    e = rel(got, want)
    assert e < tol, "forward rel err %.3e" % e
    dout = torch.randn(want.shape, generator=g, dtype=torch.float64)
    want.backward(dout)
    got.backward(dout.float().to(dev))
    for k in cpu:
        if cpu[k].requires_grad:
            assert gpu[k].grad is not None, "no grad for " + k
            e = rel(gpu[k].grad, cpu[k].grad)
            assert e < tol, "grad %s rel err %.3e" % (k, e)


def lens_mask(lens, T):
    return (torch.arange(T)[None, :] < lens[:, None])


@pytest.mark.parametrize("M,N,K", [(150, 70, 72), (300, 260, 200), (128, 128, 32), (64, 22, 96), (257, 2, 64), (5, 300, 1000)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear(dev, M, N, K, act):
    from vilco_amd import ops
    torch.manual_seed(1)
    x, w, b = torch.randn(2, M, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N)
    lens = torch.tensor([M, M - 7], dtype=torch.int32)
    m = lens_mask(lens, M)[..., None]

    def ref(x, w, b):
        z = x @ w.t() + b
        z = [z, torch.relu(z), F.gelu(z)][act]
        return z * m

    def hip(x, w, b):
        return ops.linear(x, w, b, act, lens.to(dev), M)
    run_pair(hip, ref, dict(x=x, w=w, b=b), dev, TOL_GEMM)


@pytest.mark.parametrize("span_log2,bar", [(12, 2e-5), (20, 3e-4), (26, 1e-3)])
def test_gemm_rows_of_very_different_magnitude(dev, span_log2, bar):
    """f16x2 has ONE power-of-two scale per tensor (pack.h f16_scale_from: max|x| -> [2^14, 2^15)): an element at
    2^-s of the tensor maximum keeps its second fp16 part only while that part is a normal fp16 number.  Rows of A
    spread over 2^span_log2 in magnitude: the PER-ROW relative error of A W^T stays at the 22-bit level up to a
    spread of ~2^12, below 3e-4 up to 2^20 and below the 1e-3 parity bar up to 2^26 (beyond that: `split3`, whose
    parts carry their own exponents)."""
    from vilco_amd import ops
    torch.manual_seed(5)
    M, N, K = 512, 256, 1024
    scale = torch.pow(2.0, -torch.linspace(0, span_log2, M))[:, None]
    a, w = torch.randn(M, K) * scale, torch.randn(N, K) / math.sqrt(K)
    got = ops.linear(a.to(dev), w.to(dev)).cpu().double()
    want = a.double() @ w.double().t()
    per_row = ((got - want).abs().amax(dim=1) / want.abs().amax(dim=1))
    assert float(per_row.max()) < bar, (span_log2, float(per_row.max()), int(per_row.argmax()))
    ops.set_precision("split3")                          # three bf16 parts, no shared scale: rows are independent
    try:
        got3 = ops.linear(a.to(dev), w.to(dev)).cpu().double()
    finally:
        ops.set_precision(None)
    assert float(((got3 - want).abs().amax(dim=1) / want.abs().amax(dim=1)).max()) < 2e-5


def test_range_check_finds_wide_gradients_and_switches_format(dev, monkeypatch):
    """VERDICT r04 (weak 3): the 1e10x outlier row of config W's channel attention is a DATA property; the model marks the one
    site it knows (ChannelAttention.wide_range).  ops.range_check (debug mode, VILCO_RANGE_CHECK) finds such gradient tensors
    at ANY Linear: rows of dY spread over 2^36 -- per-row errors of dX far beyond 1e-3 in the ambient fp16 x2 format, flagged
    in "warn" mode, and within the bar once "auto" reroutes that call's backward products to bf16 x3."""
    import warnings
    from vilco_amd import ops
    torch.manual_seed(9)
    M, N, K = 256, 192, 160
    x, w = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K)
    dy = torch.randn(M, N)
    dy[3] *= 2.0 ** 36                                        # one row far above the rest (the "first padded row")
    want_dx = dy.double() @ w.double()
    res = {}
    for mode in ("0", "warn", "auto"):
        monkeypatch.setattr(ops, "range_check", mode)
        ops.range_events.clear()
        ops._range_warned.clear()
        xg = x.to(dev).requires_grad_(True)
        wg = w.to(dev).requires_grad_(True)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            ops.linear(xg, wg).backward(dy.to(dev))
        per_row = ((xg.grad.cpu().double() - want_dx).abs().amax(dim=1) / want_dx.abs().amax(dim=1))
        res[mode] = (float(per_row.max()), list(ops.range_events), [str(c.message) for c in caught if "row-amax spread" in str(c.message)])
    assert res["0"][0] > 1e-2 and res["0"][1] == [] and res["0"][2] == []            # silent and wrong on the small rows
    assert res["warn"][0] > 1e-2 and len(res["warn"][1]) == 1 and len(res["warn"][2]) == 1
    assert res["warn"][1][0][:3] == (M, N, K) and res["warn"][1][0][3] > 2.0 ** 30
    assert res["auto"][0] < 1e-4 and len(res["auto"][1]) == 1, res["auto"]
    # a tensor inside the format's range is left alone
    monkeypatch.setattr(ops, "range_check", "auto")
    ops.range_events.clear()
    xg = x.to(dev).requires_grad_(True)
    ops.linear(xg, w.to(dev).requires_grad_(True)).backward(torch.randn(M, N).to(dev))
    assert ops.range_events == []


@pytest.mark.parametrize("M,N,K", [(4608, 1024, 1024), (600, 320, 256), (154, 1024, 1024)])
def test_linear_group_is_bitwise_the_individual_layers(dev, M, N, K):
    """ops.linear_group / vilco_gemm_group (round 5): the q / k / v projections of an attention block as ONE grouped launch each
    way (forward, dX).  The grouped kernel runs the same body per product: outputs and every gradient bit for bit those of
    three ops.linear calls -- at the P shape (192-row tiles, grouped), at a small shape, and at a shape whose plan splits K
    (M = 154: the library falls back to three launches)."""
    from vilco_amd import ops
    torch.manual_seed(3)
    xs = [torch.randn(2, M // 2, K) for _ in range(3)]
    ws = [torch.randn(N, K, 1) / math.sqrt(K) for _ in range(3)]
    bs = [torch.randn(N) for _ in range(3)]
    dys = [torch.randn(2, M // 2, N).to(dev) for _ in range(3)]
    res = {}
    for grouped in (False, True):
        X = [x.to(dev).requires_grad_(True) for x in xs]
        W = [w.to(dev).requires_grad_(True) for w in ws]
        Bs = [b.to(dev).requires_grad_(True) for b in bs]
        if grouped:
            Y = ops.linear_group(X, W, Bs)
        else:
            Y = [ops.linear(x, w, b) for x, w, b in zip(X, W, Bs)]
        torch.autograd.backward(Y, dys)
        res[grouped] = ([y.detach().clone() for y in Y], [t.grad.clone() for t in X + W + Bs])
    for a, b in zip(res[True][0] + res[True][1], res[False][0] + res[False][1]):
        assert torch.equal(a, b)
    want = xs[1].double() @ ws[1].double().squeeze(-1).t() + bs[1].double()
    assert rel(res[True][0][1].cpu(), want.float()) < TOL_GEMM
    # the outputs carry their max|y| partials for the pack of the next product, like ops.linear's
    Y = ops.linear_group([x.to(dev) for x in xs], [w.to(dev) for w in ws], [b.to(dev) for b in bs])
    for y in Y:
        parts, n = ops._amax_of(y)
        if parts is not None:
            assert abs(float(parts[:n].max()) - float(y.abs().max())) == 0.0
    # unequal shapes: plain per-layer path
    Y2 = ops.linear_group([xs[0].to(dev), xs[1][:, :7].contiguous().to(dev)], [w.to(dev) for w in ws[:2]], [b.to(dev) for b in bs[:2]])
    assert torch.equal(Y2[0], res[False][0][0]) and Y2[1].shape[1] == 7


@pytest.mark.parametrize("M,N,K,k3", [(4608, 2048, 256, False), (2304, 4096, 128, False), (3300, 1024, 192, False), (4600, 1024, 64, True)])
def test_gemm_tail_round_of_128_row_tiles_is_bitwise(dev, M, N, K, k3):
    """Round 5 (vilco_gemm_set_tail128): a two-part product of 257..512 192-row tiles runs as one full round of 192-row tiles over
    its first rows + one round of 128-row tiles over the rest.  A tile's arithmetic does not depend on its height: outputs and
    the input gradient (the same split on the dX product) bit for bit those of the single launch -- at 384 tiles (the split
    applies), at a ragged M, through the k = 3 conv path (tap image operand), and checked against float64."""
    from vilco_amd import ops, _lib
    lib = _lib.load()
    torch.manual_seed(5)
    x = torch.randn(2, M // 2, K)
    w = torch.randn(N, K, 3 if k3 else 1) / math.sqrt(K * (3 if k3 else 1))
    b = torch.randn(N)
    dy = torch.randn(2, M // 2, N).to(dev)
    lens = torch.tensor([M // 2, M // 2 - 5], dtype=torch.int32).to(dev)
    res = {}
    try:
        for on in (0, 1):
            _lib.check(lib.vilco_gemm_set_tail128(on))
            X, W, Bp = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
            y = ops.conv3(X, W, Bp, lens) if k3 else ops.linear(X, W, Bp)
            y.backward(dy)
            res[on] = (y.detach().clone(), X.grad.clone(), W.grad.clone(), Bp.grad.clone())
    finally:
        _lib.check(lib.vilco_gemm_set_tail128(0))      # (the library's default: measured, no gain in the step)
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)
    if not k3:
        want = x.double() @ w.double().squeeze(-1).t() + b.double()
        assert rel(res[1][0].cpu(), want.float()) < TOL_GEMM


def test_linear_unaligned_k(dev):
    from vilco_amd import ops
    x, w = torch.randn(37, 50), torch.randn(30, 50)
    run_pair(lambda x, w: ops.linear(x, w), lambda x, w: x @ w.t(), dict(x=x, w=w), dev, TOL_GEMM)


def test_linear_kn(dev):
    from vilco_amd import ops
    x, w, b = torch.randn(3, 70, 64), torch.randn(64, 4, 24) / 8, torch.randn(4, 24)
    run_pair(lambda x, w, b: ops.linear_kn(x, w, b),
             lambda x, w, b: x @ w.reshape(64, 96) + b.reshape(96), dict(x=x, w=w, b=b), dev, TOL_GEMM)


@pytest.mark.parametrize("mode,tol", [("bf16", 2e-2), ("split", 2e-4), ("split3", 2e-6), ("f16x2", 4e-6)])
def test_precision_modes(dev, mode, tol):
    from vilco_amd import ops
    x, w = torch.randn(200, 256), torch.randn(160, 256) / 16
    ops.set_precision(mode)
    try:
        run_pair(lambda x, w: ops.linear(x, w), lambda x, w: x @ w.t(), dict(x=x, w=w), dev, tol)
    finally:
        ops.set_precision(None)


@pytest.mark.parametrize("sx,sw", [(1e-9, 1e6), (3e7, 1e-12), (1.0, 1.0)])
def test_f16x2_scaling(dev, sx, sw):
    """The fp16 x2 format rescales every operand by a per-tensor power of two: results must not depend on the
    operand magnitudes (gradients ~1e-9, fp16 would flush them) and rows far below the tensor max stay accurate."""
    from vilco_amd import ops
    torch.manual_seed(5)
    x, w = torch.randn(300, 200) * sx, torch.randn(96, 200) * sw
    x[7] *= 1e-4      # a row 2^-13 below the rest
    x[9] = 0
    ops.set_precision("f16x2")
    try:
        y = ops.linear(x.to(dev), w.to(dev)).cpu().double()
    finally:
        ops.set_precision(None)
    ref = x.double() @ w.double().t()
    row_scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    err = ((y - ref).abs() / row_scale)
    err[9] = (y[9] - ref[9]).abs()
    assert float(err.max()) < 4e-6, float(err.max())
    assert float(y[9].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["f16x2", "split3", "bf16"])
@pytest.mark.parametrize("M,N,K", [(300, 260, 200), (37, 22, 50), (512, 128, 96), (129, 8, 33), (1000, 1024, 64)])
def test_packed_operands_match_unpacked(dev, mode, M, N, K):
    """One vilco_pack per tensor, consumed k-contiguous (forward), k-major as B (dX = dY W) and k-major as A and B
    (dW = dY^T X), must give bit-identical results to the calls that pack their own operands."""
    from vilco_amd import ops
    torch.manual_seed(11)
    x, w, dy = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
    ops.set_precision(mode)
    try:
        px, pw, pdy = ops.pack(x, M, K), ops.pack(w, N, K), ops.pack(dy, M, N)
        outs = []
        for planes in (False, True):
            y, dx, dw = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
            ops.gemm(x, w, y, M, N, K, 1, 1, K, K, N, a_planes=px if planes else None, b_planes=pw if planes else None)
            ops.gemm(dy, w, dx, M, K, N, 1, 0, N, K, K, a_planes=pdy if planes else None, b_planes=pw if planes else None)
            ops.gemm(dy, x, dw, N, K, M, 0, 0, N, K, K, a_planes=pdy if planes else None, b_planes=px if planes else None)
            outs.append((y, dx, dw))
        tol = {"f16x2": 4e-6, "split3": 4e-6, "bf16": 2e-2}[mode]
        for got, want in zip(outs[1], (x.double() @ w.double().t(), dy.double() @ w.double(), dy.double().t() @ x.double())):
            assert rel(got, want) < tol
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    finally:
        ops.set_precision(None)


def test_packed_operand_errors(dev):
    from vilco_amd import ops, _lib
    x, w, y = torch.randn(64, 32, device=dev), torch.randn(16, 32, device=dev), torch.empty(64, 16, device=dev)
    px = ops.pack(x, 64, 32)
    with pytest.raises(RuntimeError):      # a tapped operand's planes must be the per-sequence image (ops.pack_tap), not natural ones
        ops.gemm(x, w, y, 64, 16, 96, 1, 1, 32, 96, 16, tap=ops.TAP_A, tapC=32, tapT=64, a_planes=px)
    with pytest.raises(RuntimeError):      # ... and a plain product does not take that image
        ops.gemm(x, w, y, 64, 16, 32, 1, 1, 32, 32, 16, a_planes=ops.pack_tap(x.view(1, 64, 32)), planes_seq=(True, False))
    lib = _lib.load()
    small = torch.empty(16, dtype=torch.uint8, device=dev)
    assert lib.vilco_pack(x.data_ptr(), 64, 32, 32, 3, small.data_ptr(), 16, None) != 0   # buffer too small


def test_pack_reuse_toggle(dev, monkeypatch):
    """linear() with shared packs (default) == linear() packing per GEMM, bit for bit (forward and gradients)."""
    from vilco_amd import ops
    torch.manual_seed(12)
    res = []
    for reuse in (True, False):
        monkeypatch.setattr(ops, "_reuse_packs", reuse)
        x = torch.randn(2, 150, 72, device=dev, requires_grad=True)
        w = (torch.randn(70, 72, device=dev) / 8).requires_grad_(True)
        b = torch.randn(70, device=dev, requires_grad=True)
        torch.manual_seed(13)
        y = ops.linear(x, w, b, ops.ACT_GELU)
        y.backward(torch.randn_like(y))
        res.append((y.detach(), x.grad, w.grad, b.grad))
        torch.manual_seed(12)
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,T,Cin,Cout", [(2, 64, 96, 64), (2, 160, 64, 24), (1, 300, 32, 136), (3, 16, 8, 2)])
def test_conv3(dev, B, T, Cin, Cout):
    from vilco_amd import ops
    torch.manual_seed(2)
    x, w, b = torch.randn(B, T, Cin), torch.randn(Cout, Cin, 3) / math.sqrt(3 * Cin), torch.randn(Cout)
    lens = torch.tensor([T, max(1, T - 5), max(1, T // 2)][:B], dtype=torch.int32)
    m = lens_mask(lens, T)[..., None]

    def ref(x, w, b):
        y = F.conv1d(x.transpose(1, 2), w, b, padding=1).transpose(1, 2)
        return y * m
    run_pair(lambda x, w, b: ops.conv3(x, w, b, lens.to(dev)), ref, dict(x=x, w=w, b=b), dev, TOL_GEMM)


@pytest.mark.parametrize("C", [64, 256, 768, 1024, 2304])
@pytest.mark.parametrize("relu", [False, True])
def test_layernorm(dev, C, relu):
    from vilco_amd import ops
    torch.manual_seed(3)
    x = torch.randn(3, 37, C) * 2 + 0.5
    g, b = torch.randn(1, C, 1), torch.randn(1, C, 1)

    def ref(x, g, b):
        mu = x.mean(-1, keepdim=True)
        r = x - mu
        y = r / torch.sqrt((r ** 2).mean(-1, keepdim=True) + 1e-5) * g.view(C) + b.view(C)
        return torch.relu(y) if relu else y
    run_pair(lambda x, g, b: ops.layernorm(x, g, b, 1e-5, relu), ref, dict(x=x, g=g, b=b), dev, TOL_EW)


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("B,T,C", [(2, 64, 64), (3, 30, 128), (1, 2, 8)])
def test_dwconv3(dev, stride, B, T, C):
    from vilco_amd import ops
    torch.manual_seed(4)
    x, w = torch.randn(B, T, C), torch.randn(C, 1, 3)
    lens = torch.tensor([T, max(1, T - 3), max(1, T // 2 + 1)][:B], dtype=torch.int32)
    To = T // stride
    m = (stride * torch.arange(To)[None, :] < lens[:, None])[..., None]

    def ref(x, w):
        y = F.conv1d(x.transpose(1, 2), w, None, stride=stride, padding=1, groups=C).transpose(1, 2)
        return y * m
    run_pair(lambda x, w: ops.dwconv3(x, w, lens.to(dev), stride), ref, dict(x=x, w=w), dev, TOL_EW)


@pytest.mark.parametrize("B,T,C", [(2, 64, 64), (3, 30, 12), (1, 2, 8)])
def test_maxpool(dev, B, T, C):
    from vilco_amd import ops
    torch.manual_seed(5)
    x = torch.randn(B, T, C)
    lens = torch.tensor([T, max(1, T - 3), max(1, T // 2 + 1)][:B], dtype=torch.int32)
    m = (2 * torch.arange(T // 2)[None, :] < lens[:, None])[..., None]

    def ref(x):
        return F.max_pool1d(x.transpose(1, 2), 3, 2, 1).transpose(1, 2) * m
    run_pair(lambda x: ops.maxpool3s2(x, lens.to(dev)), ref, dict(x=x), dev, TOL_EW)


@pytest.mark.parametrize("C", [64, 22])
def test_scale_add_axpby_pe(dev, C):
    from vilco_amd import ops
    torch.manual_seed(6)
    B, T = 3, 20
    a, b = torch.randn(B, T, C), torch.randn(B, T, C)
    cs, rs = torch.randn(1, C, 1), torch.rand(B) + 0.5
    lens = torch.tensor([20, 13, 1], dtype=torch.int32)
    m = lens_mask(lens, T)[..., None]

    run_pair(lambda a, b, cs: ops.scale_add(a, b, cs, rs.to(dev), lens.to(dev), True),
             lambda a, b, cs: a * m + cs.view(C) * rs.double().view(B, 1, 1) * b, dict(a=a, b=b, cs=cs), dev, TOL_EW)
    run_pair(lambda a, b: ops.scale_add(a, b), lambda a, b: a + b, dict(a=a, b=b), dev, TOL_EW)
    run_pair(lambda a, b: ops.axpby(a, b, 0.8, 0.2), lambda a, b: 0.8 * a + 0.2 * b, dict(a=a, b=b), dev, TOL_EW)
    pe = torch.randn(T, C)
    run_pair(lambda a: ops.add_pe(a, pe.to(dev), lens.to(dev)), lambda a: a + pe.double() * m, dict(a=a), dev, TOL_EW)


def test_transpose(dev):
    from vilco_amd import ops
    x = torch.randn(3, 70, 45)
    run_pair(lambda x: ops.transpose(x), lambda x: x.transpose(1, 2), dict(x=x), dev, 1e-7)


def _attn_ref(q, k, v, lens, H, scale, xl=False):
    B, Tq, C = q.shape
    Tk = k.shape[1]
    hd = C // H
    qh = q.view(B, Tq, H, hd).transpose(1, 2)
    kh = k.view(B, Tk, H, hd).transpose(1, 2)
    vh = v.view(B, Tk, H, hd).transpose(1, 2)
    s = (qh * scale) @ kh.transpose(-1, -2)
    km = lens_mask(lens, Tk)[:, None, None, :]
    s = s.masked_fill(~km, float('-inf'))
    p = torch.softmax(s, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Tq, C)


@pytest.mark.parametrize("flash", [True, False])
@pytest.mark.parametrize("B,Tq,Tk,H,hd", [(2, 64, 64, 4, 16), (2, 40, 77, 4, 16), (1, 130, 130, 2, 64), (2, 32, 5, 4, 8),
                                          (2, 200, 157, 3, 32), (1, 96, 300, 2, 64), (1, 64, 64, 2, 128), (2, 200, 157, 2, 128),
                                          (1, 130, 300, 1, 96), (2, 77, 77, 3, 72),
                                          (2, 200, 157, 2, 144), (1, 130, 300, 1, 160), (2, 70, 70, 2, 132)])      # 129..160: the 160-wide tiles (config W: 144)
def test_attention(dev, B, Tq, Tk, H, hd, flash):
    from vilco_amd import ops
    ops.use_flash = flash
    torch.manual_seed(7)
    C = H * hd
    q, k, v = torch.randn(B, Tq, C), torch.randn(B, Tk, C), torch.randn(B, Tk, C)
    lens = torch.tensor([Tk, max(1, Tk - 9)][:B], dtype=torch.int32)
    scale = 1 / math.sqrt(hd)
    try:
        run_pair(lambda q, k, v: ops.attention(q, k, v, lens.to(dev), H, scale),
                 lambda q, k, v: _attn_ref(q, k, v, lens, H, scale), dict(q=q, k=k, v=v), dev, TOL_GEMM)
    finally:
        ops.use_flash = True


@pytest.mark.parametrize("mode,tol", [("split3", TOL_GEMM), ("f16x2", TOL_GEMM), ("split", 5e-4)])
def test_attention_precision_modes(dev, mode, tol):
    from vilco_amd import ops
    torch.manual_seed(8)
    B, Tq, Tk, H, hd = 2, 200, 157, 3, 32
    q, k, v = torch.randn(B, Tq, H * hd) * 3, torch.randn(B, Tk, H * hd) * 1e-3, torch.randn(B, Tk, H * hd) * 50
    lens = torch.tensor([Tk, Tk - 9], dtype=torch.int32)
    ops.set_precision(mode)
    try:
        run_pair(lambda q, k, v: ops.attention(q, k, v, lens.to(dev), H, 0.2),
                 lambda q, k, v: _attn_ref(q, k, v, lens, H, 0.2), dict(q=q, k=k, v=v), dev, tol)
    finally:
        ops.set_precision(None)


@pytest.mark.parametrize("flash,T,hd", [(True, 48, 16), (False, 48, 16), (True, 100, 64), (False, 100, 64),
                                         (True, 576, 64), (True, 160, 64), (True, 100, 128)])     # 576: several key tiles, band GEMM,
# dS as operand planes from the dQ kernel (round 5; 160: a partial last key tile on that path; 100: the fp32 dS + relshift pack path)
def test_rel_attention(dev, flash, T, hd):
    """XLNet core vs the published formula incl. rel_shift_bnij (modeling_xlnet_x.py:256-320)."""
    from vilco_amd import ops
    ops.use_flash = flash
    torch.manual_seed(8)
    B, H = 2, 4
    C = H * hd
    qw, qr, k, v = [torch.randn(B, T, C) for _ in range(4)]
    kr = torch.randn(2 * T, C)
    lens = torch.tensor([T, T - 11], dtype=torch.int32)
    scale = 1 / math.sqrt(hd)

    def ref(qw, qr, k, v, kr):
        f = lambda x: x.view(B, T, H, hd).permute(1, 0, 2, 3)          # ibnd
        ac = torch.einsum("ibnd,jbnd->bnij", f(qw), f(k))
        krr = kr.view(2 * T, 1, H, hd).expand(2 * T, B, H, hd)
        bd = torch.einsum("ibnd,jbnd->bnij", f(qr), krr)
        xs = bd.shape
        bd = bd.reshape(xs[0], xs[1], xs[3], xs[2])[:, :, 1:, :].reshape(xs[0], xs[1], xs[2], xs[3] - 1)[:, :, :, :T]
        score = (ac + bd) * scale
        pad = (~lens_mask(lens, T)).double()                          # [B, T(j)]
        mask = ((pad[:, None, None, :] - torch.eye(T, dtype=torch.float64)[None, None]) > 0).double()
        score = score - 1e30 * mask
        p = torch.softmax(score, dim=3)
        o = torch.einsum("bnij,jbnd->ibnd", p, f(v))
        return o.permute(1, 0, 2, 3).reshape(B, T, C)
    try:
        run_pair(lambda qw, qr, k, v, kr: ops.rel_attention(qw, qr, k, v, kr, lens.to(dev), H, scale), ref,
                 dict(qw=qw, qr=qr, k=k, v=v, kr=kr), dev, TOL_GEMM)
    finally:
        ops.use_flash = True


def test_channel_attention(dev):
    from vilco_amd import ops
    torch.manual_seed(9)
    B, T, H, hd = 2, 50, 4, 16
    C = H * hd
    qkv = torch.randn(B, T, 3 * C)
    scale = hd ** -0.5

    def ref(qkv):
        x = qkv.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = x[0], x[1], x[2]
        att = ((k * scale).transpose(-1, -2) @ v).softmax(dim=-1)
        o = (att @ q.transpose(-1, -2)).transpose(-1, -2)
        return o.transpose(1, 2).reshape(B, T, C)
    run_pair(lambda qkv: ops.channel_attention(qkv, H, scale), ref, dict(qkv=qkv), dev, TOL_GEMM)


def test_dropout_op(dev):
    from vilco_amd import ops
    torch.manual_seed(31)
    x = torch.randn(3, 50, 40, device=dev, requires_grad=True)
    ops.dropout_log = []
    try:
        y = ops.dropout(x, 0.25, True, "t")
        (site, p, seed, shape), = ops.dropout_log
    finally:
        ops.dropout_log = None
    m = ops.dropout_mask(p, seed, shape, dev)
    assert torch.equal(y, x * m)
    assert abs(float((m == 0).float().mean()) - 0.25) < 0.03
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.equal(x.grad, g * m)
    assert ops.dropout(x, 0.25, False) is x and ops.dropout(x, 0.0, True) is x
    y2 = ops.dropout(x, 0.25, True)                      # a new call draws a new mask
    assert not torch.equal(y2, y)


def test_attention_dropout_mask_statistics(dev):
    """Round 5: the attention-probability mask has a function of its own (one strong hash per mask row + a two-multiply finalizer
    per element, csrc/common.h).  Drop rate, independence of neighbours along both axes and across seeds, binomial spread of the
    per-row / per-column rates -- what a Bernoulli mask has and a careless cheap hash does not (the first candidate had a
    correlation of -0.0036 between elements 64 columns apart)."""
    from vilco_amd import ops
    p, shape = 0.1, (2, 4, 768, 768)
    d = (ops.dropout_mask(p, 1234, shape, dev, "attn_prob") == 0).float().reshape(-1, shape[-1])
    d2 = (ops.dropout_mask(p, 1235, shape, dev, "attn_prob") == 0).float().reshape(-1, shape[-1])
    keep = ops.dropout_mask(p, 1234, shape, dev, "attn_prob")
    assert set(keep.unique().tolist()) <= {0.0, float(torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))}
    rate = float(d.mean())
    n = d.numel()
    assert abs(rate - p) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4, rate
    x, y = d - rate, d2 - rate
    var = rate * (1 - rate)

    def corr(a, b):
        return float((a * b).mean()) / var
    tol = 5.0 / math.sqrt(n)                       # five standard errors of a sample correlation
    for lag in (1, 2, 3, 4, 16, 64, 128, 256):
        assert abs(corr(x[:, :-lag], x[:, lag:])) < tol, ("columns", lag)
        assert abs(corr(x[:-lag], x[lag:])) < tol, ("rows", lag)
    assert abs(corr(x, y)) < tol                   # another seed: another mask
    for axis, m in ((1, shape[-1]), (0, d.shape[0])):
        sd = float(d.mean(axis).std())
        assert 0.85 < sd / math.sqrt(var / m) < 1.15, (axis, sd)


@pytest.mark.parametrize("T,hd", [(100, 16), (130, 64), (130, 128), (130, 144)])
def test_attention_prob_dropout(dev, T, hd):
    """attention with dropout on the probabilities == reference attention with the same mask (fwd + grads)."""
    from vilco_amd import ops
    torch.manual_seed(32)
    B, H = 2, 2
    C = H * hd
    q, k, v = [torch.randn(B, T, C, device=dev, requires_grad=True) for _ in range(3)]
    lens = torch.tensor([T, T - 9], dtype=torch.int32, device=dev)
    ops.dropout_log = []
    try:
        o = ops.attention(q, k, v, lens, H, 0.3, drop_p=0.2)
        (site, p, seed, shape), = ops.dropout_log
    finally:
        ops.dropout_log = None
    assert shape == (B, H, T, T) and site == "attn_prob"
    m = ops.dropout_mask(p, seed, shape, dev, site).double().cpu()
    g = torch.randn(B, T, C)
    o.backward(g.to(dev))
    qd, kd, vd = [t.detach().double().cpu().requires_grad_(True) for t in (q, k, v)]
    qh, kh, vh = [t.view(B, T, H, hd).transpose(1, 2) for t in (qd, kd, vd)]
    s = (qh * 0.3) @ kh.transpose(-2, -1)
    km = (torch.arange(T)[None, :] < lens.cpu()[:, None])[:, None, None, :]
    pr = torch.softmax(s.masked_fill(~km, float('-inf')), dim=-1) * m
    want = (pr @ vh).transpose(1, 2).reshape(B, T, C)
    want.backward(g.double())
    assert rel(o, want) < TOL_GEMM
    for a, b in ((q, qd), (k, kd), (v, vd)):
        assert rel(a.grad, b.grad) < TOL_GEMM


def test_attention_zero_upstream_gradient(dev):
    """dO == 0 for the whole tensor (a clip dropped by stochastic depth): the fp16x2 operand scale of dO is then 2^126;
    every gradient must come out exactly zero, not inf * 0."""
    from vilco_amd import ops
    torch.manual_seed(33)
    B, T, H, hd = 2, 96, 2, 64
    q, k, v = [torch.randn(B, T, H * hd, device=dev, requires_grad=True) for _ in range(3)]
    lens = torch.tensor([T, T - 5], dtype=torch.int32, device=dev)
    o = ops.attention(q, k, v, lens, H, 0.125)
    o.backward(torch.zeros_like(o))
    for t in (q, k, v):
        assert torch.count_nonzero(t.grad) == 0
    x, w = torch.zeros(64, 32, device=dev, requires_grad=True), torch.randn(16, 32, device=dev, requires_grad=True)
    y = ops.linear(x, w)
    y.backward(torch.zeros_like(y))
    assert torch.count_nonzero(y) == 0 and torch.count_nonzero(w.grad) == 0 and torch.count_nonzero(x.grad) == 0


def test_attention_randomized_shapes(dev):
    """48 seeded random problems (ragged lengths, head dims 4..128 that are multiples of 4, cross-attention shapes,
    with / without probability dropout, both fp32-equivalent precisions) against an fp64 reference: forward and all
    three gradients."""
    from vilco_amd import ops
    rng = np.random.RandomState(1234)
    for case in range(48):
        B, H = int(rng.randint(1, 4)), int(rng.randint(1, 5))
        hd = int(rng.choice([4, 8, 12, 16, 24, 32, 40, 48, 64])) if case < 40 else int(rng.choice([68, 96, 100, 128]))
        Tq, Tk = int(rng.randint(1, 300)), int(rng.randint(1, 300))
        p = float(rng.choice([0.0, 0.0, 0.15, 0.5]))
        prec = str(rng.choice(["f16x2", "f16x2", "split3"]))
        if hd > 64:
            prec = "f16x2"                      # bf16 x3 above hd 64 has no fused kernels (LDS), hence no dropout
        torch.manual_seed(1000 + case)
        C = H * hd
        q = torch.randn(B, Tq, C, device=dev, requires_grad=True)
        k, v = [torch.randn(B, Tk, C, device=dev, requires_grad=True) for _ in range(2)]
        lens = torch.tensor(rng.randint(1, Tk + 1, size=B), dtype=torch.int32, device=dev)
        scale = float(rng.uniform(0.05, 0.5))
        ops.set_precision(prec)
        ops.dropout_log = []
        try:
            o = ops.attention(q, k, v, lens, H, scale, drop_p=p)
            log = list(ops.dropout_log)
        finally:
            ops.dropout_log = None
            ops.set_precision(None)
        m = ops.dropout_mask(log[0][1], log[0][2], log[0][3], dev, log[0][0]).double().cpu() if log else 1.0
        g = torch.randn(B, Tq, C)
        o.backward(g.to(dev))
        qd, kd, vd = [t.detach().double().cpu().requires_grad_(True) for t in (q, k, v)]
        qh = qd.view(B, Tq, H, hd).transpose(1, 2)
        kh, vh = [t.view(B, Tk, H, hd).transpose(1, 2) for t in (kd, vd)]
        sc = (qh * scale) @ kh.transpose(-2, -1)
        km = (torch.arange(Tk)[None, :] < lens.cpu()[:, None])[:, None, None, :]
        want = ((torch.softmax(sc.masked_fill(~km, float('-inf')), dim=-1) * m) @ vh).transpose(1, 2).reshape(B, Tq, C)
        want.backward(g.double())
        tag = (case, B, H, hd, Tq, Tk, p, prec)
        assert rel(o, want) < TOL_GEMM, tag
        for a, b in ((q, qd), (k, kd), (v, vd)):
            assert rel(a.grad, b.grad) < TOL_GEMM, tag


def test_gemm_randomized_shapes(dev):
    """60 seeded random products in the three orientations the model uses (forward, dX, dW), with and without shared
    packs, odd sizes included, against fp64."""
    from vilco_amd import ops
    rng = np.random.RandomState(4321)
    for case in range(60):
        M, N, K = [int(x) for x in rng.randint(1, 700, size=3)]
        torch.manual_seed(2000 + case)
        x, w, dy = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
        use_planes = bool(rng.randint(0, 2))
        px, pw, pdy = (ops.pack(x, M, K), ops.pack(w, N, K), ops.pack(dy, M, N)) if use_planes else (None, None, None)
        y, dx, dw = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
        ops.gemm(x, w, y, M, N, K, 1, 1, K, K, N, a_planes=px, b_planes=pw)
        ops.gemm(dy, w, dx, M, K, N, 1, 0, N, K, K, a_planes=pdy, b_planes=pw)
        ops.gemm(dy, x, dw, N, K, M, 0, 0, N, K, K, a_planes=pdy, b_planes=px)
        for got, want in ((y, x.double() @ w.double().t()), (dx, dy.double() @ w.double()), (dw, dy.double().t() @ x.double())):
            assert rel(got, want) < 4e-6, (case, M, N, K, use_planes)


@pytest.mark.parametrize("act", [0, 2])
@pytest.mark.parametrize("M,N,K", [(150, 72, 64), (700, 130, 96)])
def test_linear_fused_dropout(dev, act, M, N, K):
    """nn.Dropout after a linear layer, fused into the GEMM epilogue (forward) and into act_bwd (backward), equals the
    unfused composition with the same mask."""
    from vilco_amd import ops
    torch.manual_seed(41)
    x = torch.randn(2, M, K, device=dev, requires_grad=True)
    w = (torch.randn(N, K, device=dev) / 8).requires_grad_(True)
    b = torch.randn(N, device=dev, requires_grad=True)
    lens = torch.tensor([M, M - 7], dtype=torch.int32, device=dev)
    ops.dropout_log = []
    try:
        y = ops.linear(x, w, b, act, lens, M, drop_p=0.3, drop_site="t")
        (site, p, seed, shape), = ops.dropout_log
    finally:
        ops.dropout_log = None
    assert shape == (2, M, N)
    m = ops.dropout_mask(p, seed, shape, dev)
    g = torch.randn_like(y)
    y.backward(g)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    for t in (x, w, b):
        t.grad = None
    y2 = ops.linear(x, w, b, act, lens, M) * m
    y2.backward(g)
    for a, c in zip(got, (y2.detach(), x.grad, w.grad, b.grad)):
        assert rel(a, c) < 2e-6


@pytest.mark.parametrize("B,Tq,Tk,H,lens", [(2, 128, 128, 2, [128, 1]), (1, 129, 257, 1, [200]), (3, 31, 77, 2, [77, 64, 5]),
                                            (2, 640, 640, 4, [640, 333]), (1, 2, 3, 1, [3])])
def test_attention_hd64_fast_path(dev, B, Tq, Tk, H, lens):
    """the hd = 64 kernels (attn_fwd64 / attn_bwd_dq64 / attn_bwd_dkdv64: 128-query workgroups, P and dS in registers,
    transposing LDS reads, amax partials of every output) at tile-edge shapes: query counts that are not multiples of
    128 / 32, key counts that are not multiples of 64, a single valid key, cross-attention shapes"""
    from vilco_amd import ops
    torch.manual_seed(17)
    C = H * 64
    q, k, v = torch.randn(B, Tq, C) * 1.5, torch.randn(B, Tk, C) * 0.7, torch.randn(B, Tk, C) * 20
    lt = torch.tensor(lens, dtype=torch.int32)
    run_pair(lambda q, k, v: ops.attention(q, k, v, lt.to(dev), H, 0.125),
             lambda q, k, v: _attn_ref(q, k, v, lt, H, 0.125), dict(q=q, k=k, v=v), dev, TOL_GEMM)
    # the partials the kernels leave for the next operand pack equal the true maxima
    qd, kd, vd = [t.to(dev).requires_grad_(True) * 1.0 for t in (q, k, v)]      # non-leaf: a hook sees the gradient tensor itself
    got = {}
    for name, t in (("q", qd), ("k", kd), ("v", vd)):
        t.register_hook(lambda g, name=name: got.__setitem__(name, g))
    o = ops.attention(qd, kd, vd, lt.to(dev), H, 0.125)
    parts, n = ops._amax_of(o)
    assert parts is not None and abs(float(parts[:n].max()) - float(o.abs().max())) == 0.0
    o.backward(torch.randn_like(o))
    for name in ("q", "k", "v"):
        parts, n = ops._amax_of(got[name])
        assert parts is not None and abs(float(parts[:n].max()) - float(got[name].abs().max())) == 0.0, name


@pytest.mark.parametrize("M,N,K,act", [(4608, 1024, 1024, 0), (300, 200, 96, 2), (154, 1024, 4096, 0), (2304, 4096, 1024, 2)])
def test_gemm_output_amax_partials(dev, M, N, K, act):
    """vilco_gemm_desc.amax_out: the partial maxima the MFMA kernel's epilogue (or the split-K reduce) leaves equal
    max|C| of the stored output exactly, for tiled, ragged and split-K plans, with and without an activation"""
    from vilco_amd import ops
    torch.manual_seed(M + N)
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    pre = torch.empty_like(C) if act == 2 else None
    ops.gemm(A, B, C, M, N, K, 1, 1, K, K, N, bias=bias, preact=pre, act=act, want_amax=True)
    parts, n = ops._amax_of(C)
    assert parts is not None and n > 0
    assert float(parts[:n].max()) == float(C.abs().max())
    # an in-place edit invalidates the tag (and the remembered operand planes)
    planes = ops.pack(C, M, N)
    assert ops.pack(C, M, N) is planes
    C.mul_(2.0)
    assert ops._amax_of(C)[0] is None and ops.pack(C, M, N) is not planes


@pytest.mark.parametrize("M,N,K", [(288, 1024, 768), (154, 1024, 768), (144, 1024, 512), (576, 512, 512), (640, 512, 512), (16, 512, 512),
                                   (17, 100, 100), (33, 64, 64), (161, 130, 96), (320, 72, 700), (288, 1024, 1024)])
@pytest.mark.parametrize("act", [0, 2])
def test_few_row_kernel_against_tiled_kernel_and_float64(dev, M, N, K, act):
    """Round 6 (gemm_skinny_kernel, vilco_gemm_set_skinny): NT products with M <= 640 rows as ONE launch whose eight waves split K.
    Against float64, against the tiled kernels' plan for the same call (same arithmetic per product, another summation order:
    agreement to fp32 rounding of the sum, not bit for bit), with bias / GELU + pre-activation / residual / row lengths / fused
    dropout (the mask is a function of the element index: the same in both) and exact max|C| partials; ragged M, N, K."""
    from vilco_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(M * 7 + N)
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / math.sqrt(K)
    bias, resid = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    T = M // 2 if M % 2 == 0 else M
    lens = torch.tensor([T, T - 3] if M % 2 == 0 else [M - 2], dtype=torch.int32, device=dev)
    out = {}
    try:
        for on in (1, 0):
            _lib.check(lib.vilco_gemm_set_skinny(on))
            assert int(lib.vilco_gemm_config_gen()) > 0
            C = torch.empty(M, N, device=dev)
            pre = torch.empty_like(C) if act == 2 else None
            ops.gemm(A, B, C, M, N, K, 1, 1, K, K, N, bias=bias, preact=pre, act=act, residual=resid, row_len=lens, rowT=T,
                     drop=(0.25, 1234), want_amax=True)
            parts, n = ops._amax_of(C)
            assert parts is not None and n > 0 and float(parts[:n].max()) == float(C.abs().max()), on
            out[on] = (C.clone(), None if pre is None else pre.clone(), n)
    finally:
        _lib.check(lib.vilco_gemm_set_skinny(1))
    if K <= 768 and N <= 1024:      # (the shapes the plan sends to the few-row kernel: one max|C| partial per workgroup of ITS grid)
        bm = 16 if M <= int(os.environ.get("VILCO_GEMM_SKINNY_BM16", "640")) else 32      # (32-row workgroups: lab setting)
        assert out[1][2] == ((N + 63) // 64) * ((M + bm - 1) // bm)
    z = A.double() @ B.double().t() + bias.double()
    if act == 2:
        assert rel(out[1][1].cpu(), z.float().cpu()) < TOL_GEMM
        assert rel(out[1][1], out[0][1]) < 2e-6
    assert rel(out[1][0], out[0][0]) < 2e-6
    keep = out[0][0] != resid                                   # (where dropout or the row mask zeroed the product, C == residual)
    zz = (F.gelu(z) if act == 2 else z).float()
    rows = torch.arange(M, device=dev)
    valid = ((rows % T) < lens[rows // T])[:, None]
    want = torch.where(valid & keep, zz / 0.75, torch.zeros_like(zz)) + resid
    assert rel(out[1][0].cpu(), want.cpu()) < TOL_GEMM


@pytest.mark.parametrize("M,N,K,form", [(154, 1024, 1024, "nt"), (154, 1024, 4096, "nt"), (288, 1024, 1024, "nn"), (1024, 1024, 4608, "tn"),
                                        (77, 333, 2000, "nt"), (1152, 1024, 4096, "nn"), (130, 64, 640, "tn")])
def test_split_k_fixup_equals_reduce_kernel(dev, M, N, K, form):
    """split-K problems finished inside the launch (the tile's last-arriving workgroup sums the partial accumulators in
    split order: vilco_gemm_set_fixup(1)) against the fp32-slab + splitk_reduce_kernel form: bit for bit,
    with bias / activation / row mask / amax partials in the epilogue, and repeatable (the counters reset themselves)."""
    from vilco_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(M + K)
    if form == "nt":
        A, B, a_kc, b_kc, lda, ldb = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), 1, 1, K, K
        want = A.double() @ B.double().t()
    elif form == "nn":
        A, B, a_kc, b_kc, lda, ldb = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), 1, 0, K, N
        want = A.double() @ B.double()
    else:
        A, B, a_kc, b_kc, lda, ldb = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), 0, 0, M, N
        want = A.double().t() @ B.double()
    bias = torch.randn(N, device=dev)
    lens = torch.tensor([M - 3], dtype=torch.int32, device=dev)
    outs = []
    try:
        for mode in (0, 1, 1, 1):
            _lib.check(lib.vilco_gemm_set_fixup(mode))
            C = torch.full((M, N), float('nan'), device=dev)
            pre = torch.empty_like(C)
            ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, bias=bias, preact=pre, act=2, row_len=lens, rowT=M, want_amax=True)
            parts, n = ops._amax_of(C)
            outs.append((C, pre, float(parts[:n].max()) if parts is not None else None))
    finally:
        _lib.check(lib.vilco_gemm_set_fixup(0))
    ref = torch.nn.functional.gelu(want + bias.double())
    ref[M - 3:] = 0
    assert rel(outs[0][0], ref) < 4e-6
    for C, pre, am in outs[1:]:
        assert torch.equal(C, outs[0][0]) and torch.equal(pre, outs[0][1])
        assert am == float(C.abs().max()) == outs[0][2]


@pytest.mark.parametrize("M,N,K,form", [(1024, 1024, 4608, "tn"), (1024, 4096, 4608, "tn"), (300, 200, 96, "tn"), (257, 130, 2080, "nt"),
                                        (512, 384, 160, "nn"), (1024, 1024, 9082, "tn")])
def test_single_part_products_two_k_steps_per_interval(dev, M, N, K, form):
    """precision 4 (the leading fp16 part of both operands, one MFMA per product; weight gradients with long
    contractions): the kernel covers two K-steps per barrier interval (gemm.hip: K2) -- odd step counts, split-K plans,
    all three operand orientations, against the product of the fp16-rounded operands in float64."""
    from vilco_amd import ops
    torch.manual_seed(M + N + K)
    if form == "nt":
        A, B, a_kc, b_kc, lda, ldb = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), 1, 1, K, K
        Ad, Bd = A, B.t()
    elif form == "nn":
        A, B, a_kc, b_kc, lda, ldb = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), 1, 0, K, N
        Ad, Bd = A, B
    else:
        A, B, a_kc, b_kc, lda, ldb = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), 0, 0, M, N
        Ad, Bd = A.t(), B

    def lead(t):            # what the kernel multiplies: fp16(x * s) / s with s the power of two that puts max|x| in [2^14, 2^15)
        s = 2.0 ** (14 - torch.floor(torch.log2(t.abs().max())).item())
        return (t * s).half().double() / s
    want = lead(Ad) @ lead(Bd)
    C = torch.empty(M, N, device=dev)
    ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, precision=4)
    assert rel(C, want) < 2e-6, rel(C, want)
    # packed operands (what the weight-gradient products of the model use)
    pa, pb = ops.pack(A, A.shape[0], A.shape[1]), ops.pack(B, B.shape[0], B.shape[1])
    C2 = torch.empty(M, N, device=dev)
    ops.gemm(A, B, C2, M, N, K, a_kc, b_kc, lda, ldb, N, precision=4, a_planes=pa, b_planes=pb)
    assert rel(C2, want) < 2e-6


@pytest.mark.parametrize("rows,C,act,p,masked", [(4608, 1024, 2, 0.1, True), (100, 96, 0, 0.0, True), (333, 64, 1, 0.3, False), (77, 1024, 2, 0.0, False)])
def test_act_bwd_writes_operand_planes(dev, rows, C, act, p, masked):
    """vilco_act_bwd_planes: dz written by the activation-backward kernel as fp16 x2 operand planes (scale from a bound on
    max|dz| instead of the exact maximum) -- the planes decode to the fp32 dz of the plain kernel to 22 bits, their zero rows
    are zero, the bias gradient is unchanged, and both backward products read them like the planes `pack` makes."""
    from vilco_amd import ops, _lib
    torch.manual_seed(rows + C)
    T = rows
    dy = torch.randn(rows, C, device=dev) * 3e-4
    dy[5, :] *= 40.0
    aux = torch.randn(rows, C, device=dev)
    lens = torch.tensor([rows - 9], dtype=torch.int32, device=dev) if masked else None
    drop = (p, 1234)
    dz_ref, db_ref = ops._act_bwd(dy, aux if act else None, act, lens, T if masked else None, True, drop)
    parts = torch.stack([dy[: rows // 2].abs().max(), dy[rows // 2:].abs().max()])          # what the producer of dy would leave
    ops._tag_amax(dy, parts, 2)
    dz, db, planes = ops._act_bwd(dy, aux if act else None, act, lens, T if masked else None, True, drop, planes=True)
    assert dz is None and planes is not None and planes.numel() == _lib.load().vilco_pack_bytes(rows, C, 3)
    assert torch.equal(db, db_ref)
    hdr = planes[:4096 + 512].view(torch.float32)
    inv_s, s = float(hdr[1024]), float(hdr[1025])
    assert inv_s * s == 1.0
    bound = float(dy.abs().max()) / (1.0 - p) * (1.13 if act == 2 else 1.0)
    assert 2.0 ** 14 <= bound * s < 2.0 ** 15
    rows32 = (rows + 31) // 32 * 32
    body = planes[4096 + 512:].view(torch.float16).view(2, rows32, C).double()
    dec = (body[0] + body[1]) / s
    assert float(dec[rows:].abs().max()) == 0.0 if rows32 > rows else True
    err = (dec[:rows] - dz_ref.double()).abs()
    assert bool((err <= dz_ref.double().abs() * 2.0 ** -21 + bound * 2.0 ** -39).all())
    # the two products of a Linear backward on these planes vs on pack(dz)
    N2 = 72
    w = torch.randn(C, N2, device=dev) / 8          # dX = dZ W (NN), dW = dZ^T X (TN)
    x = torch.randn(rows, N2, device=dev)
    pw, px, pz = ops.pack(w, C, N2), ops.pack(x, rows, N2), ops.pack(dz_ref, rows, C)
    for pl in (planes, pz):
        dx, dw = torch.empty(rows, N2, device=dev), torch.empty(C, N2, device=dev)
        ops.gemm(None, w, dx, rows, N2, C, 1, 0, C, N2, N2, a_planes=pl, b_planes=pw)
        ops.gemm(None, x, dw, C, N2, rows, 0, 0, C, N2, N2, a_planes=pl, b_planes=px)
        assert rel(dx, dz_ref.double() @ w.double()) < 4e-6 and rel(dw, dz_ref.double().t() @ x.double()) < 4e-6


def test_linear_backward_takes_producer_planes(dev, monkeypatch):
    """a Linear whose upstream gradient carries amax partials (here: the dX of the layer above) runs its backward without a pack
    of dz and without an fp32 dz; gradients equal the VILCO_PRODUCER_PLANES=0 path to operand precision"""
    from vilco_amd import ops
    torch.manual_seed(3)
    x = torch.randn(2, 300, 64, device=dev, requires_grad=True)
    w1 = (torch.randn(256, 64, device=dev) / 8).requires_grad_(True)
    b1 = torch.randn(256, device=dev, requires_grad=True)
    w2 = (torch.randn(64, 256, device=dev) / 16).requires_grad_(True)
    lens = torch.tensor([300, 250], dtype=torch.int32, device=dev)
    g = torch.randn(2, 300, 64, device=dev)
    calls = []
    real = ops._act_bwd

    def spy(*a, **k):              # (flags only: a held reference to db would make AccumulateGrad copy it before its deferred finish)
        out = real(*a, **k)
        calls.append(tuple(t is not None for t in out))
        return out
    monkeypatch.setattr(ops, "_act_bwd", spy)
    res = []
    for on in (True, False):
        monkeypatch.setattr(ops, "producer_planes", on)
        for t in (x, w1, b1, w2):
            t.grad = None
        calls.clear()
        y = ops.linear(ops.linear(x, w1, b1, ops.ACT_GELU, lens, 300), w2, None, ops.ACT_NONE, lens, 300)
        y.backward(g)
        res.append([t.grad.clone() for t in (x, w1, b1, w2)])
        fc1 = [c for c in calls if len(c) == 3 and c[1]]          # the GELU layer's call (it has the bias): (dz?, db?, planes?)
        assert fc1 == [(False, True, True) if on else (True, True, False)]
    for a, c in zip(*res):
        assert rel(a, c) < 2e-6


def test_scale_add_backward_tags_branch_gradient(dev):
    """the branch gradient scale_add's backward writes carries its exact max|db| partials (vilco_scale_add_bwd_amax)"""
    from vilco_amd import ops
    torch.manual_seed(8)
    B, T, C = 2, 50, 96
    a = torch.randn(B, T, C, device=dev, requires_grad=True)
    b = torch.randn(B, T, C, device=dev, requires_grad=True) * 1.0
    cs = torch.randn(1, C, 1, device=dev, requires_grad=True)
    lens = torch.tensor([50, 31], dtype=torch.int32, device=dev)
    got = {}
    b.register_hook(lambda g: got.__setitem__("db", g))
    ops.scale_add(a, b, cs, None, lens, True).backward(torch.randn(B, T, C, device=dev))
    parts, n = ops._amax_of(got["db"])
    assert parts is not None and n > 0 and float(parts[:n].max()) == float(got["db"].abs().max())


@pytest.mark.parametrize("B,T,C,relu,layout", [(2, 2304, 1024, False, "nat"), (3, 37, 96, False, "nat"), (2, 50, 64, True, "seq"),
                                               (2, 4541, 1024, True, "seq"), (1, 77, 1000, False, "nat"), (2, 40, 72, True, "seq")])
def test_layernorm_writes_operand_planes(dev, B, T, C, relu, layout):
    """vilco_layernorm_fwd_planes: the LayerNorm kernel also writes its output as fp16 x2 operand planes, scaled by the bound
    max|gamma| sqrt(C) + max|beta| -- natural rows (a Linear's input) or the k=3 convs' zero-padded per-sequence image.  The
    planes decode to y to 22 bits, every padding row is zero, the consumer's pack finds them, and the products on them equal
    the products on packed planes."""
    from vilco_amd import ops
    torch.manual_seed(B * T + C)
    x = torch.randn(B, T, C, device=dev) * 3 + 1
    g = (torch.randn(C, device=dev) * 0.5 + 1).requires_grad_(True)
    bt = (torch.randn(C, device=dev) * 0.2).requires_grad_(True)
    y_ref = ops.layernorm(x, g, bt, 1e-5, relu).detach()
    y = ops.layernorm(x, g, bt, 1e-5, relu, planes=layout)
    assert torch.equal(y.detach(), y_ref)
    hit = getattr(y, "_vilco_planes" if layout == "nat" else "_vilco_tap_planes", None)
    if (C % 32 if layout == "nat" else C % 8):
        assert hit is None          # unsupported width: the hint is ignored, the consumer packs
        return
    planes = hit[0]
    assert (ops.pack(y, B * T, C) if layout == "nat" else ops.pack_tap(y)) is planes          # what the consumer will find
    hdr = planes[:4096 + 512].view(torch.float32)
    inv_s, s = float(hdr[1024]), float(hdr[1025])
    bound = float(g.detach().abs().max()) * math.sqrt(C) + float(bt.detach().abs().max())
    assert inv_s * s == 1.0 and 2.0 ** 14 <= bound * s < 2.0 ** 15 and float(y_ref.abs().max()) <= bound
    if layout == "nat":
        rows32 = (B * T + 31) // 32 * 32
        body = planes[4096 + 512:].view(torch.float16).view(2, rows32, C).double()
        dec = (body[0] + body[1]) / s
        want = torch.zeros(rows32, C, dtype=torch.float64, device=dev)
        want[:B * T] = y_ref.view(B * T, C).double()
    else:
        tr = (B * (T + 2) + 31) // 32 * 32 + 64
        ps = (tr * C + 7) // 8 * 8
        raw = planes[4096 + 512:].view(torch.float16)
        dec = ((raw[:tr * C].double() + raw[ps:ps + tr * C].double()) / s).view(tr, C)
        want = torch.zeros(tr, C, dtype=torch.float64, device=dev)
        for b in range(B):
            want[b * (T + 2) + 1: b * (T + 2) + 1 + T] = y_ref[b].double()
    assert bool(((dec - want).abs() <= want.abs() * 2.0 ** -21 + bound * 2.0 ** -39).all())
    pad = dec[want.abs().sum(1) == 0]
    assert pad.numel() == 0 or float(pad.abs().max()) == 0.0          # zero rows are exactly zero
    # the consumer: a Linear / a k=3 conv forward + backward on the producer's planes vs on its own pack
    N = 40
    w = (torch.randn(N, C, device=dev) / 8) if layout == "nat" else (torch.randn(N, C, 3, device=dev) / 8)
    res = []
    for hint in (layout, None):
        xx = x.clone().requires_grad_(True)
        ww = w.clone().requires_grad_(True)
        yy = ops.layernorm(xx, g, bt, 1e-5, relu, planes=hint)
        out = ops.linear(yy, ww) if layout == "nat" else ops.conv3(yy, ww)
        out.backward(torch.ones_like(out) / 7)
        res.append((out.detach(), xx.grad, ww.grad))
    for a, c in zip(*res):
        assert rel(a, c) < 2e-6


def test_layernorm_row_mask_equals_masking_afterwards(dev):
    """LayerNorm + ReLU with the LevelCat separator rows zeroed inside the kernel (row_mask) == the op followed by the mask
    multiply, forward and backward, and the operand image written alongside holds the masked rows"""
    from vilco_amd import ops
    torch.manual_seed(12)
    B, T, C = 2, 45, 64
    mask = (torch.rand(T, device=dev) > 0.2).float()[None, :, None].contiguous()
    g0, b0 = torch.randn(C, device=dev) * 0.5 + 1, torch.randn(C, device=dev) * 0.3
    x0, dy = torch.randn(B, T, C, device=dev), torch.randn(B, T, C, device=dev)
    res = []
    for fused in (True, False):
        x, g, b = x0.clone().requires_grad_(True), g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = ops.layernorm(x, g, b, 1e-5, True, "seq", mask) if fused else ops.layernorm(x, g, b, 1e-5, True) * mask
        if fused:
            planes = ops.pack_tap(y)
            assert planes is y._vilco_tap_planes[0]
            parts, n = ops._amax_of(y)
            assert float(parts[:n].max()) == float(y.abs().max())
        y.backward(dy)
        res.append((y.detach(), x.grad, g.grad, b.grad))
    for a, c in zip(*res):
        assert torch.equal(a, c) or rel(a, c) < 1e-6
    assert float((res[0][0] * (1 - mask)).abs().max()) == 0.0



def test_conv3_row_mask_equals_masking_afterwards(dev):
    """vilco_gemm_desc.row_mask / vilco_act_bwd_planes(row_mask): a k=3 conv whose output rows are zeroed by a per-row 0 / 1 mask in
    the GEMM epilogue == the conv followed by the multiply, forward and all three gradients"""
    from vilco_amd import ops
    torch.manual_seed(21)
    B, T, Cin, Cout = 2, 70, 64, 48
    mask = (torch.rand(B, T, 1, device=dev) > 0.3).float().contiguous()
    x0, w0, b0 = torch.randn(B, T, Cin, device=dev), torch.randn(Cout, Cin, 3, device=dev) / 8, torch.randn(Cout, device=dev)
    dy = torch.randn(B, T, Cout, device=dev)
    res = []
    for fused in (True, False):
        x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = ops.conv3(x, w, b, None, row_mask=mask) if fused else ops.conv3(x, w, b, None) * mask
        y.backward(dy)
        res.append((y.detach(), x.grad, w.grad, b.grad))
    assert torch.equal(res[0][0], res[1][0])
    for a, c in zip(*res):
        assert rel(a, c) < 2e-6


def test_bias_add_matches_the_plain_expression(dev):
    """ops.bias_add (round 6): x + b with the bias gradient summed by vilco_colsum instead of ATen's multi-block `sum` (which does
    not replay inside a hipGraph on the null stream with this torch / ROCm: tools/lab/sum_graph_probe.py)"""
    from vilco_amd import ops
    torch.manual_seed(2)
    x0, b0, dy = torch.randn(2, 300, 128, device=dev), torch.randn(4, 32, device=dev), torch.randn(2, 300, 128, device=dev)
    res = []
    for ours in (True, False):
        x, b = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = ops.bias_add(x, b) if ours else x + b.view(1, 1, -1)
        y.backward(dy)
        res.append((y.detach(), x.grad, b.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][2].shape == b0.shape and rel(res[0][2], res[1][2]) < 1e-6


def test_pack_group_equals_single_packs_and_keeps_the_cache(dev):
    """ops.pack_group (round 6): three tensors of one shape through ONE vilco_pack_many launch -- the same plane bytes as three
    ops.pack calls, each tensor tagged so that a later pack is a cache hit; a tensor that already carries its planes keeps them."""
    from vilco_amd import ops
    torch.manual_seed(3)
    xs = [torch.randn(300, 96, device=dev) * (i + 1) for i in range(3)]
    ys = [x.clone() for x in xs]
    got = ops.pack_group(xs, 300, 96)
    want = [ops.pack(y, 300, 96) for y in ys]
    hdr = 4096 + 512                                       # (amax partials + {1/s, s}: only the scale pair and the planes are defined bytes)
    for g, w in zip(got, want):
        assert torch.equal(g[hdr:], w[hdr:]) and torch.equal(g[4096:4104], w[4096:4104])
    for x, g in zip(xs, got):
        assert ops.pack(x, 300, 96) is g                   # remembered on the tensor
    again = ops.pack_group(xs, 300, 96)                     # every tensor is a hit now: nothing is re-packed
    assert all(a is g for a, g in zip(again, got))
    zs = [torch.randn(300, 96, device=dev) for _ in range(3)]
    first = ops.pack(zs[1], 300, 96)
    mixed = ops.pack_group(zs, 300, 96)                     # one hit among them: per-tensor packs, the hit stays a hit
    assert mixed[1] is first and torch.equal(mixed[0][hdr:], ops.pack(zs[0].clone(), 300, 96)[hdr:])


@pytest.mark.parametrize("masked_by", ["row_mask", "lens"])
def test_conv3_backward_writes_dz_image_from_the_mask_kernel(dev, masked_by):
    """Round 6 (vilco_act_bwd_planes_seq + vilco_layernorm_bwd_res_amax): conv k=3 -> LayerNorm -> ReLU, the heads' / embeddings'
    pattern (MQ/libs/modeling/meta_archs.py:216-235).  LayerNorm backward leaves max|dx| partials, so the conv's mask kernel
    writes dZ straight into the zero-padded operand image of the dX and weight-gradient products (no fp32 dZ, no pack_tap): the
    decoded image equals the masked gradient to 22 bits with zero pad rows, and all gradients equal the VILCO_CONV_DZ_PLANES=0
    path (same products on planes whose scale comes from a bound instead of the exact maximum)."""
    from vilco_amd import _lib, ops
    torch.manual_seed(5)
    B, T, Cin, Cout = 2, 70, 64, 96
    x0, w0 = torch.randn(B, T, Cin, device=dev), torch.randn(Cout, Cin, 3, device=dev) / 8
    g0, b0 = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
    dy = torch.randn(B, T, Cout, device=dev)
    mask = (torch.rand(B, T, 1, device=dev) > 0.3).float().contiguous() if masked_by == "row_mask" else None
    lens = torch.tensor([T, T - 23], dtype=torch.int32, device=dev) if masked_by == "lens" else None
    res, saved = [], ops.conv_dz_planes
    try:
        for direct in (True, False):
            ops.conv_dz_planes = direct
            x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
            g, b = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            z = ops.conv3(x, w, None, lens, row_mask=mask)
            y = ops.layernorm(z, g, b, 1e-5, True)
            y.backward(dy)
            res.append((x.grad, w.grad, g.grad, b.grad))
    finally:
        ops.conv_dz_planes = saved
    for a, c in zip(*res):
        assert rel(a, c) < 2e-6
    # the image itself: LayerNorm backward's dx, tagged, through the mask kernel alone
    dz_in = torch.randn(B, T, Cout, device=dev)
    parts = torch.zeros(ops.AMAX_PARTS, device=dev)
    parts[0] = dz_in.abs().max()
    ops._tag_amax(dz_in, parts, 1)
    dz_none, _, pz = ops._act_bwd(dz_in, None, ops.ACT_NONE, lens, T, False, planes="seq", row_mask=mask)
    assert dz_none is None and pz is not None
    hdr = pz[:4096 + 512].view(torch.float32)
    inv_s, s = float(hdr[1024]), float(hdr[1025])
    assert inv_s * s == 1.0
    rows_out = (B * (T + 2) + 31) // 32 * 32 + 64
    stride = (rows_out * Cout + 7) // 8 * 8
    body = pz[4096 + 512:].view(torch.float16)
    dec = ((body[:rows_out * Cout].double() + body[stride:stride + rows_out * Cout].double()) / s).view(rows_out, Cout)
    keep = mask.view(B, T, 1) if mask is not None else (torch.arange(T, device=dev)[None, :, None] < lens[:, None, None]).float()
    want = torch.zeros(rows_out, Cout, dtype=torch.float64, device=dev)
    for bi in range(B):
        want[bi * (T + 2) + 1: bi * (T + 2) + 1 + T] = (dz_in[bi] * keep[bi]).double()
    top = float(dz_in.abs().max())
    assert bool(((dec - want).abs() <= want.abs() * 2.0 ** -21 + top * 2.0 ** -37).all())


@pytest.mark.parametrize("B,Tq,Tk,H,lens", [(2, 300, 300, 2, [300, 211]), (1, 129, 77, 4, [77]), (2, 2304, 2304, 2, [2304, 1500])])
def test_attention_writes_output_planes(dev, B, Tq, Tk, H, lens):
    """vilco_attn_fwd_planes (hd = 64 forward kernels): the attention output also leaves the kernel as the fp16 x2 operand planes
    of the output projection, scaled by max|v| (a convex combination of rows of v cannot exceed it): decode == o to 22 bits,
    zero rows zero, the projection finds the planes, and its forward + backward equal the VILCO_ATTN_PLANES=0 path"""
    from vilco_amd import ops
    torch.manual_seed(B * Tq + H)
    C = H * 64
    q, k = torch.randn(B, Tq, C, device=dev), torch.randn(B, Tk, C, device=dev)
    v = torch.randn(B, Tk, C, device=dev) * 3
    lt = torch.tensor(lens, dtype=torch.int32, device=dev)
    o = ops.attention(q, k, v, lt, H, 0.125)
    hit = getattr(o, "_vilco_planes", None)
    assert hit is not None and ops.pack(o, B * Tq, C) is hit[0]
    planes = hit[0]
    hdr = planes[:4096 + 512].view(torch.float32)
    inv_s, s = float(hdr[1024]), float(hdr[1025])
    vmax = float(v.abs().max())
    assert inv_s * s == 1.0 and 2.0 ** 14 <= vmax * s < 2.0 ** 15 and float(o.abs().max()) <= vmax * (1 + 1e-5)
    rows32 = (B * Tq + 31) // 32 * 32
    body = planes[4096 + 512:].view(torch.float16).view(2, rows32, C).double()
    dec = (body[0] + body[1]) / s
    want = torch.zeros(rows32, C, dtype=torch.float64, device=dev)
    want[:B * Tq] = o.view(B * Tq, C).double()
    assert bool(((dec - want).abs() <= want.abs() * 2.0 ** -21 + vmax * 2.0 ** -39).all())
    w = torch.randn(C, C, device=dev) / 8
    res = []
    for on in (True, False):
        ops.attn_planes = on
        try:
            qq, kk, vv, ww = [t.clone().requires_grad_(True) for t in (q, k, v, w)]
            oo = ops.attention(qq, kk, vv, lt, H, 0.125)
            assert (getattr(oo, "_vilco_planes", None) is not None) == on
            y = ops.linear(oo, ww)
            y.backward(torch.ones_like(y) / 5)
            res.append((y.detach(), qq.grad, vv.grad, ww.grad))
        finally:
            ops.attn_planes = True
    for a, c in zip(*res):
        assert rel(a, c) < 2e-6
