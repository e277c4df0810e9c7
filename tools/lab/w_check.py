import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import torch
import vilco_amd.modeling as vm
from oracle import mq_oracle as O
from vilco_amd import ops
from vilco_amd.core.config import make_config
from parity_util import rel_err
import cases
dev = torch.device("cuda:0")
import bench
for (D, H, flash, xl, T) in [(256, 2, True, False, 64)]:
    ops.use_flash = flash
    over = cases.overrides(D=D, T=T, Cin=384, Ctxt=768, H=H, use_xl=xl, droppath=0.1)
    cfg = make_config(**over)['model']
    torch.manual_seed(0)
    kw = dict(cfg, xlnet_config=dict(bench.P_XLNET, d_model=D, n_head=H, d_head=D // H, d_inner=512, dropout=0.0)) if xl else cfg
    model = vm.make_meta_arch('LocPointTransformer', **kw)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if 'drop_path' in n_:
                p_.fill_(0.3)
    model.eval()
    vl = cases.video_list(T, 384, 768, 77)
    print('   clip lengths:', [int(v['feats'].shape[-1]) for v in vl])
    p64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in model.state_dict().items()}
    vl64 = [{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()} for d in vl]
    want, _ = O.forward_losses(p64, cfg, vl64)
    want['final_loss'].backward()
    model = model.to(dev)
    model.loss_normalizer = cfg['train_cfg']['init_loss_norm']
    tgt = model.backbone.stem[0].channel_attn.attn.qkv.weight
    orig_gemm = ops.gemm
    def spy(A, B, Cc, M, N, K, *a, **k):
        r = orig_gemm(A, B, Cc, M, N, K, *a, **k)
        if Cc.shape == tgt.shape and M == tgt.shape[0] and N == tgt.shape[1]:
            print('   dW-shaped product M=%d N=%d K=%d: max|C| %.3e' % (M, N, K, float(Cc.abs().max())), flush=True)
        return r
    ops.gemm = spy
    ob = ops._ChannelAttn.backward
    def bspy(ctx, dout):
        r = ob(ctx, dout)
        g = r[0]; Cn = g.shape[-1] // 3
        i_ = int(g[..., :Cn].abs().argmax()); T_ = g.shape[1]
        print('   dout max %.2e | dq argmax (b,t,c)=(%d,%d,%d) | rows with |dq|>1e-3: %s' % (float(dout.abs().max()), i_ // (T_ * Cn), (i_ // Cn) % T_, i_ % Cn, sorted(set(((g[..., :Cn].abs() > 1e-3).nonzero()[:, 1]).tolist()))[:12]), flush=True)
        for i, nm in enumerate('qkv'):
            sl = g[..., i * Cn:(i + 1) * Cn].abs()
            print('   d%s: max %.2e  median %.2e  q99 %.2e' % (nm, float(sl.max()), float(sl.median()), float(sl.flatten().kthvalue(int(sl.numel() * 0.99)).values)), flush=True)
        return r
    ops._ChannelAttn.backward = staticmethod(bspy)
    losses = model(vl, is_training=True)
    losses['final_loss'].backward()
    ops.gemm = orig_gemm
    print('   accumulated grad max %.3e' % float(tgt.grad.abs().max()))
    errs = sorted(((rel_err(p.grad, p64[k].grad, 1e-7), k, float(p64[k].grad.abs().max())) for k, p in model.named_parameters() if p64[k].grad is not None and p.grad is not None), reverse=True)
    print("D=%d H=%d T=%d xl=%s loss err %.2e | worst:" % (D, H, T, xl, rel_err(losses['final_loss'], want['final_loss'])), ["%.1e %s (max|g| %.1e)" % e for e in errs[:4]], flush=True)
