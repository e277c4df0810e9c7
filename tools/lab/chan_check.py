"""ChannelAttention core at realistic widths vs float64: where does the qkv-weight gradient error of a twice-applied stem block come from?"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from vilco_amd import ops
from parity_util import rel_err
dev = torch.device("cuda:0")
def ref(qkv, B, T, H, hd, scale):
    C = H * hd
    x = qkv.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    att = ((k * scale).transpose(-1, -2) @ v).softmax(dim=-1)
    o = (att @ q.transpose(-1, -2)).transpose(-1, -2)
    return o.transpose(1, 2).reshape(B, T, C)
for (B, T, H, hd, amp) in [(2, 64, 2, 128, 1.0), (2, 256, 2, 128, 1.0), (2, 256, 2, 128, 3.0), (2, 64, 2, 128, 3.0), (2, 256, 2, 144, 3.0), (2, 256, 8, 32, 3.0)]:
    torch.manual_seed(1)
    C = H * hd
    qkv = (torch.randn(B, T, 3 * C) * amp)
    scale = hd ** -0.5
    a = qkv.double().requires_grad_(True)
    w = torch.randn(B, T, C, dtype=torch.float64)
    o = ref(a, B, T, H, hd, scale); (o * w).sum().backward()
    g = qkv.to(dev).requires_grad_(True)
    og = ops.channel_attention(g, H, scale); (og * w.float().to(dev)).sum().backward()
    dq, dk, dv = [rel_err(g.grad[..., i * C:(i + 1) * C], a.grad[..., i * C:(i + 1) * C]) for i in range(3)]
    print("B=%d T=%d H=%d hd=%d amp=%.0f: out %.1e  dq %.1e dk %.1e dv %.1e   max|logit| %.1f" % (B, T, H, hd, amp, rel_err(og, o), dq, dk, dv,
          float(((a.detach().reshape(B, T, 3, H, hd)[:, :, 1] * scale).transpose(1, 2).transpose(-1, -2) @ a.detach().reshape(B, T, 3, H, hd)[:, :, 2].transpose(1, 2)).abs().max())), flush=True)
