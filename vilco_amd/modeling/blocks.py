"""Drop-in operators of the MQ model (reference: MQ/libs/modeling/blocks.py), computed by the HIP
kernels in libvilco_hip.so.

Same class names, constructor arguments and state_dict keys/shapes as the reference, so reference
checkpoints load unchanged.  Two call forms per module:

  forward(x[B,C,T], mask[B,1,T]) -> the reference's channel-first signature (transposes at the edge)
  forward_tm(x[B,T,C], lens[B])  -> token-major fast path used inside the backbone

Masks are prefix masks (meta_archs.py:1175), carried as int32 valid lengths.  nn.Conv1d /
nn.Linear / nn.LayerNorm members are parameter containers only (identical init + key names +
the isinstance checks of make_optimizer, train_utils.py:76-77); their forward is never called.
"""
import math

import numpy as np
import os

import torch
from torch import nn

from .. import ops
from ..ops import ACT_GELU, ACT_NONE


# ------------------------------------------------------------------------------ layout helpers
def mask_to_lens(mask):
    """bool/float mask [B,1,T] or [B,T] (prefix form) -> int32 lengths [B]."""
    B = mask.shape[0]
    return mask.reshape(B, -1).to(torch.int32).sum(dim=1, dtype=torch.int32)


def lens_to_mask(lens, T):
    """int32 lengths [B] -> bool mask [B,1,T] (meta_archs.py:1175,1179)."""
    return (torch.arange(T, device=lens.device)[None, :] < lens[:, None]).unsqueeze(1)


def to_tm(x):
    """[B,C,T] -> token-major [B,T,C]."""
    return ops.transpose(x.contiguous())


def from_tm(x):
    return ops.transpose(x.contiguous())


def down_lens(lens, stride):
    """nearest-neighbour mask[::stride] (blocks.py:116-122): valid(t') = stride*t' < len."""
    if stride == 1:
        return lens
    return torch.div(lens + (stride - 1), stride, rounding_mode="floor").to(torch.int32)


_DROP_POOL = {}          # (device, B, keep) -> [rows of ready factors, next row]
_DROP_POOL_ROWS = 64


def reset_drop_pool():
    """forget the drawn stochastic-depth factors.  Called around a stream capture (vilco_amd/graph.py): the pool drawn
    inside the capture is re-drawn by every replay (torch's graph-safe Philox offset), rows drawn before it would be
    replayed as constants."""
    _DROP_POOL.clear()


def _drop_rowscale(x, drop_prob, training):
    """per-sample stochastic-depth factor (blocks.py:628-641) as a [B] row scale, or None.
    The reference draws `keep + rand(B)`, floors it and divides by keep at each of its ~30 sites per step (four tiny
    launches each); the same factors are drawn here 64 sites at a time and handed out row by row."""
    if drop_prob == 0.0 or not training:
        return None
    keep = 1.0 - drop_prob
    B = x.shape[0]
    key = (x.device, B, keep, torch.cuda.current_stream().cuda_stream if x.is_cuda else 0)     # one pool per stream
    pool = _DROP_POOL.get(key)
    if pool is None or pool[1] >= _DROP_POOL_ROWS:
        r = torch.rand(_DROP_POOL_ROWS, B, dtype=torch.float32, device=x.device)
        pool = _DROP_POOL[key] = [torch.floor(keep + r).div_(keep), 0]
    rs = pool[0][pool[1]]
    pool[1] += 1
    if ops.dropout_log is not None:          # parity tests replay the factors into the oracle
        ops.dropout_log.append(("droppath", rs))
    return rs


# ------------------------------------------------------------------------------ MaskedConv1D
class MaskedConv1D(nn.Module):
    """Masked 1-D conv (blocks.py:57-130).  Supported on the path: dense k=3 s=1, dense k=1,
    depthwise k=3 s in {1,2} without bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode='zeros'):
        super().__init__()
        assert (kernel_size % 2 == 1) and (kernel_size // 2 == padding)
        self.stride = stride
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding, dilation,
                              groups, bias, padding_mode)
        if bias:
            torch.nn.init.constant_(self.conv.bias, 0.)

    def augment_classification(self, num_new_classes, device):
        """grow the output channels, keeping the old rows (blocks.py:85-104)."""
        old = self.conv
        n_old = old.out_channels
        new = nn.Conv1d(old.in_channels, n_old + num_new_classes, 3, stride=1, padding=1, dilation=1,
                        groups=1, bias=True, padding_mode='zeros').to(device)
        torch.nn.init.constant_(new.bias, -(math.log((1 - 0.01) / 0.01)))
        new.weight.data[:n_old] = old.weight.data[:n_old]
        new.bias.data[:n_old] = old.bias.data[:n_old]
        self.conv = new

    def forward_tm(self, x, lens):
        c = self.conv
        k, s = c.kernel_size[0], self.stride
        T = x.shape[1]
        assert T % s == 0
        out_lens = down_lens(lens, s)
        if c.groups == 1 and k == 1 and s == 1:
            y = ops.linear(x, c.weight, c.bias, ACT_NONE, out_lens, T)
        elif c.groups == 1 and k == 3 and s == 1:
            y = ops.conv3(x, c.weight, c.bias, out_lens)
        elif c.groups == c.in_channels == c.out_channels and k == 3 and c.bias is None and s in (1, 2):
            y = ops.dwconv3(x, c.weight, lens, s)
        else:
            raise NotImplementedError("MaskedConv1D variant not on the MQ hot path: k=%d s=%d groups=%d"
                                      % (k, s, c.groups))
        return y, out_lens

    def forward(self, x, mask):
        B, C, T = x.size()
        assert T % self.stride == 0
        y, out_lens = self.forward_tm(to_tm(x), mask_to_lens(mask))
        return from_tm(y), lens_to_mask(out_lens, T // self.stride)


class LayerNorm(nn.Module):
    """LayerNorm over channels of [B,C,T] (blocks.py:133-175); weight/bias keep shape [1,C,1]."""

    def __init__(self, num_channels, eps=1e-5, affine=True, device=None, dtype=None):
        super().__init__()
        kw = {'device': device, 'dtype': dtype}
        self.num_channels, self.eps, self.affine = num_channels, eps, affine
        if affine:
            self.weight = nn.Parameter(torch.ones([1, num_channels, 1], **kw))
            self.bias = nn.Parameter(torch.zeros([1, num_channels, 1], **kw))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)

    def forward_tm(self, x, relu=False, planes=None, row_mask=None, skip=False):
        """planes: "nat" / "seq" when the output goes straight into a Linear / a k=3 conv; row_mask: see ops.layernorm;
        skip: -> (y, x_skip), x_skip = x for the residual connection around the branch y feeds (ops.layernorm)"""
        assert x.shape[-1] == self.num_channels
        return ops.layernorm(x, self.weight, self.bias, self.eps, relu, planes, row_mask, skip, site=getattr(self, '_site', None))

    def forward(self, x):
        assert x.dim() == 3 and x.shape[1] == self.num_channels
        return from_tm(self.forward_tm(to_tm(x)))


def get_sinusoid_encoding(n_position, d_hid):
    """[1, C, T] sinusoid table (blocks.py:179-190), vectorised in float64 then cast."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    ang = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab = np.where(j % 2 == 0, np.sin(ang), np.cos(ang))
    return torch.FloatTensor(tab).unsqueeze(0).transpose(1, 2)


# ------------------------------------------------------------------------------ attention
class MaskedMHA(nn.Module):
    """Multi-head (cross-)attention with key mask (blocks.py:194-269)."""

    def __init__(self, n_embd, n_head, attn_pdrop=0.0, proj_pdrop=0.0):
        super().__init__()
        assert n_embd % n_head == 0
        self.n_embd, self.n_head = n_embd, n_head
        self.n_channels = n_embd // n_head
        self.scale = 1.0 / math.sqrt(self.n_channels)
        self.key = nn.Conv1d(n_embd, n_embd, 1)
        self.query = nn.Conv1d(n_embd, n_embd, 1)
        self.value = nn.Conv1d(n_embd, n_embd, 1)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.proj_drop = nn.Dropout(proj_pdrop)
        self.proj = nn.Conv1d(n_embd, n_embd, 1)

    def forward_tm(self, x, lens, enc=None, enc_lens=None):
        src, src_lens = (x, lens) if enc is None else (enc, enc_lens)
        q = ops.linear(x, self.query.weight, self.query.bias)
        k = ops.linear(src, self.key.weight, self.key.bias)
        v = ops.linear(src, self.value.weight, self.value.bias)
        o = ops.attention(q, k, v, src_lens, self.n_head, self.scale,
                          drop_p=self.attn_drop.p if self.training else 0.0)            # attn_drop (blocks.py:253)
        out = ops.linear(o, self.proj.weight, self.proj.bias, ACT_NONE, lens, x.shape[1],
                         drop_p=self.proj_drop.p if self.training else 0.0, drop_site="proj_drop")   # proj_drop (:264)
        return out, lens

    def forward(self, x, mask, encoder_hidden_states=None, encoder_attention_mask=None):
        lens = mask_to_lens(mask)
        enc = enc_lens = None
        if encoder_hidden_states is not None:
            enc, enc_lens = to_tm(encoder_hidden_states), mask_to_lens(encoder_attention_mask)
        y, _ = self.forward_tm(to_tm(x), lens, enc, enc_lens)
        return from_tm(y), mask


class MaskedMHCA(nn.Module):
    """Self-attention whose q/k/v come from depthwise conv + LayerNorm (blocks.py:272-410).
    Note the reference strides the QUERY conv with n_kv_stride too (blocks.py:313); kept."""

    def __init__(self, n_embd, n_head, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0):
        super().__init__()
        assert n_embd % n_head == 0
        self.n_embd, self.n_head = n_embd, n_head
        self.n_channels = n_embd // n_head
        self.scale = 1.0 / math.sqrt(self.n_channels)
        assert (n_qx_stride == 1) or (n_qx_stride % 2 == 0)
        assert (n_kv_stride == 1) or (n_kv_stride % 2 == 0)
        self.n_qx_stride, self.n_kv_stride = n_qx_stride, n_kv_stride
        ks = n_qx_stride + 1 if n_qx_stride > 1 else 3
        self.query_conv = MaskedConv1D(n_embd, n_embd, ks, stride=n_kv_stride, padding=ks // 2,
                                       groups=n_embd, bias=False)
        self.query_norm = LayerNorm(n_embd)
        ks = n_kv_stride + 1 if n_kv_stride > 1 else 3
        self.key_conv = MaskedConv1D(n_embd, n_embd, ks, stride=n_kv_stride, padding=ks // 2,
                                     groups=n_embd, bias=False)
        self.key_norm = LayerNorm(n_embd)
        self.value_conv = MaskedConv1D(n_embd, n_embd, ks, stride=n_kv_stride, padding=ks // 2,
                                       groups=n_embd, bias=False)
        self.value_norm = LayerNorm(n_embd)
        self.key = nn.Conv1d(n_embd, n_embd, 1)
        self.query = nn.Conv1d(n_embd, n_embd, 1)
        self.value = nn.Conv1d(n_embd, n_embd, 1)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.proj_drop = nn.Dropout(proj_pdrop)
        self.proj = nn.Conv1d(n_embd, n_embd, 1)

    def fusable(self):
        """the q/k/v pre-projection can run as ONE kernel together with the block's ln1 (ops.qkv_pre)"""
        c = self.query_conv.conv
        return (ops.use_qkv_pre and ops.qkv_pre_supported(self.n_embd) and c.kernel_size[0] == 3
                and self.query_conv.stride == self.key_conv.stride == self.value_conv.stride and self.query_conv.stride in (1, 2)
                and self.query_norm.eps == self.key_norm.eps == self.value_norm.eps and self.query_norm.affine)

    def forward_tm_fused(self, x, lens, ln1, want_h):
        """(attention output, q_lens, h = ln1(x) or None, x_skip): ln1 + the three depthwise convs + their LayerNorms in one
        launch; x_skip = x for the block's skip connection (its gradient is added inside ln1's backward kernel)"""
        s = self.query_conv.stride
        outs = ops.qkv_pre(x, (ln1.weight, ln1.bias, ln1.eps),
                           (self.query_conv.conv.weight, self.key_conv.conv.weight, self.value_conv.conv.weight),
                           ((self.query_norm.weight, self.query_norm.bias), (self.key_norm.weight, self.key_norm.bias),
                            (self.value_norm.weight, self.value_norm.bias), self.query_norm.eps), lens, s, want_h, skip=True)
        q, k, v = outs[:3]
        q_lens = down_lens(lens, s)
        out, q_lens = self._attend(q, k, v, q_lens, q_lens)
        return out, q_lens, (outs[3] if want_h else None), outs[-1]

    def forward_tm(self, x, lens):
        q, q_lens = self.query_conv.forward_tm(x, lens)
        q = self.query_norm.forward_tm(q, planes="nat")          # (each goes straight into its projection)
        k, kv_lens = self.key_conv.forward_tm(x, lens)
        k = self.key_norm.forward_tm(k, planes="nat")
        v, _ = self.value_conv.forward_tm(x, lens)
        v = self.value_norm.forward_tm(v, planes="nat")
        return self._attend(q, k, v, q_lens, kv_lens)

    def _attend(self, q, k, v, q_lens, kv_lens):
        # the three projections (blocks.py:332-344) as grouped launches when q, k, v have one shape (self-attention at equal strides)
        q, k, v = ops.linear_group([q, k, v], [self.query.weight, self.key.weight, self.value.weight],
                                   [self.query.bias, self.key.bias, self.value.bias])
        o = ops.attention(q, k, v, kv_lens, self.n_head, self.scale,
                          drop_p=self.attn_drop.p if self.training else 0.0)            # attn_drop (blocks.py:394)
        out = ops.linear(o, self.proj.weight, self.proj.bias, ACT_NONE, q_lens, q.shape[1],
                         drop_p=self.proj_drop.p if self.training else 0.0, drop_site="proj_drop")   # proj_drop (:405)
        return out, q_lens

    def forward(self, x, mask):
        T = x.shape[-1]
        y, q_lens = self.forward_tm(to_tm(x), mask_to_lens(mask))
        return from_tm(y), lens_to_mask(q_lens, T // self.n_kv_stride)


class ChannelAttention(nn.Module):
    """Attention over channels: softmax((k*scale)^T v) applied to q (blocks.py:412-436). [B,T,C] in/out."""

    # Which of this block's backward products leave the ambient operand format for bf16 x3 when `wide_range` is set
    # ("qkv", "core", "proj"; env VILCO_CA_WIDE overrides, for experiments).  The module ignores the mask (blocks.py:459-466,
    # kept), so the padded rows of a short clip carry non-zero values through it; in a backbone that applies the block
    # twice (no XLNet layer, backbones.py:276-278) the first padded row's gradient reaches ~1e10 x the typical element, more
    # exponent range than ONE power-of-two scale per tensor leaves to the other rows of the fp16 x2 planes.
    WIDE_PARTS = tuple(x for x in os.environ.get("VILCO_CA_WIDE", "qkv,core,proj").split(",") if x)

    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.wide_range = False          # set by the backbone on the block it applies twice

    def _bp(self, part):
        return 2 if (self.wide_range and part in self.WIDE_PARTS and ops.get_precision() == 3) else None

    def forward(self, x):
        qkv = ops.linear(x.contiguous(), self.qkv.weight, self.qkv.bias, bwd_precision=self._bp("qkv"))
        a = ops.channel_attention(qkv, self.num_heads, self.scale, bwd_precision=self._bp("core"))
        return ops.linear(a, self.proj.weight, self.proj.bias, bwd_precision=self._bp("proj"))


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob


class ChannelBlock(nn.Module):
    """x + attn(x), then x + mlp(norm2(x)) -- norm1 exists but is never applied and no mask is
    used (blocks.py:459-466); reproduced as is."""

    def __init__(self, n_embd, num_heads, mlp_ratio=4., qkv_bias=False, drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, ffn=True, cpe_act=False):
        super().__init__()
        self.ffn = ffn
        self.norm1 = norm_layer(n_embd)
        self.attn = ChannelAttention(n_embd, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        if self.ffn:
            self.norm2 = norm_layer(n_embd)
            n_hidden = int(n_embd * mlp_ratio)
            self.mlp = nn.Sequential(nn.Linear(n_embd, n_hidden), act_layer(), nn.Linear(n_hidden, n_embd))

    def _dp(self, x):
        p = getattr(self.drop_path, "drop_prob", 0.0) or 0.0
        return _drop_rowscale(x, p, self.training)

    def forward_tm(self, x):
        cur = self.attn(x)
        x = ops.scale_add(x, cur, None, self._dp(x))
        if self.ffn:
            h = ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, planes="nat")
            h = ops.linear(h, self.mlp[0].weight, self.mlp[0].bias, ACT_GELU)
            h = ops.linear(h, self.mlp[2].weight, self.mlp[2].bias)
            x = ops.scale_add(x, h, None, self._dp(x))
        return x

    def forward(self, x):
        return from_tm(self.forward_tm(to_tm(x)))


class Scale(nn.Module):
    """learnable scalar multiplier (blocks.py:605-623)."""

    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(init_value, dtype=torch.float32), requires_grad=True)

    def forward(self, x):
        return x * self.scale


class AffineDropPath(nn.Module):
    """per-channel scale (init 1e-4) + per-sample stochastic depth (blocks.py:655-670)."""

    def __init__(self, num_dim, drop_prob=0.0, init_scale_value=1e-4):
        super().__init__()
        self.scale = nn.Parameter(init_scale_value * torch.ones((1, num_dim, 1)), requires_grad=True)
        self.drop_prob = drop_prob

    def forward(self, x):  # channel-first reference form
        y = self.scale * x
        rs = _drop_rowscale(x, self.drop_prob, self.training)
        return y if rs is None else y * rs.view(-1, 1, 1)


class TransformerBlock(nn.Module):
    """conv-attention block with optional text cross-attention, MLP and the channel-attention mix
    (blocks.py:468-593).  ln3 and drop_path_attn are shared by the self and cross paths, as in the
    reference (:571-573)."""

    def __init__(self, n_embd, n_head, n_ds_strides=(1, 1), n_out=None, n_hidden=None,
                 act_layer=nn.GELU, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, t_c_alpha=0.8,
                 use_rel_pe=False, use_cross_modal=False, use_adaper=True):
        super().__init__()
        assert len(n_ds_strides) == 2
        self.t_c_alpha = t_c_alpha
        self.ln1 = LayerNorm(n_embd)
        self.ln2 = LayerNorm(n_embd)
        self.attn = MaskedMHCA(n_embd, n_head, n_qx_stride=n_ds_strides[0], n_kv_stride=n_ds_strides[1],
                               attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop)
        self.use_cross_modal = use_cross_modal
        if use_cross_modal:
            self.cross_attn = MaskedMHA(n_embd, n_head, attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop)
            self.ln3 = LayerNorm(n_embd)
            self.cross_pool_skip = nn.Identity()
        self.n_ds_strides = n_ds_strides
        if n_ds_strides[0] > 1:
            assert n_ds_strides[0] == 2, "only the stride-2 pyramid of the shipped configs is implemented"
            self.pool_skip = nn.MaxPool1d(n_ds_strides[0] + 1, stride=n_ds_strides[0],
                                          padding=(n_ds_strides[0] + 1) // 2)
        else:
            self.pool_skip = nn.Identity()
        if n_hidden is None:
            n_hidden = 4 * n_embd
        if n_out is None:
            n_out = n_embd
        self.mlp = nn.Sequential(nn.Conv1d(n_embd, n_hidden, 1), act_layer(),
                                 nn.Dropout(proj_pdrop, inplace=True), nn.Conv1d(n_hidden, n_out, 1),
                                 nn.Dropout(proj_pdrop, inplace=True))
        self.channel_attn = ChannelBlock(n_embd, n_head, drop_path=path_pdrop)
        if path_pdrop > 0.0:
            self.drop_path_attn = AffineDropPath(n_embd, drop_prob=path_pdrop)
            self.drop_path_mlp = AffineDropPath(n_out, drop_prob=path_pdrop)
        else:
            self.drop_path_attn = nn.Identity()
            self.drop_path_mlp = nn.Identity()
        self.use_adaper = use_adaper
        self.adapters = None   # {"attn": time-axis Adapter} attached by the meta-arch (use_adapt)

    # AdapterMixin surface used by meta_archs.attach_pets (blocks.py:27-54)
    def attach_adapter(self, **kwargs):
        if not isinstance(self.adapters, nn.ModuleDict):
            self.adapters = nn.ModuleDict()
        for name, adapter in kwargs.items():
            self.adapters[name] = adapter

    def _dp(self, mod, x):
        if isinstance(mod, AffineDropPath):
            return mod.scale, _drop_rowscale(x, mod.drop_prob, self.training)
        return None, None

    def forward_tm(self, x, lens, cross_y=None, cross_lens=None):
        need_h = (self.n_ds_strides[0] == 1 and self.n_ds_strides[1] == 1) or (self.adapters is not None and "attn" in self.adapters)
        # Every residual branch opens with a LayerNorm; its input comes back from that op as `xs` / `out_s` for the skip
        # connection, so the gradient over the skip is added inside the LayerNorm backward kernel (ops.layernorm `skip`).
        if self.attn.fusable() and self.ln1.affine:
            a, out_lens, h, xs = self.attn.forward_tm_fused(x, lens, self.ln1, need_h)
        else:
            h, xs = self.ln1.forward_tm(x, skip=True)
            a, out_lens = self.attn.forward_tm(h, lens)
        if self.adapters is not None and "attn" in self.adapters:
            a = a + self.adapters["attn"].forward_tm(h)          # parallel adapter (meta_archs.py:144-148)
        skip = ops.maxpool3s2(xs, lens) if self.n_ds_strides[0] > 1 else xs
        cs, rs = self._dp(self.drop_path_attn, a)
        out = ops.scale_add(skip, a, cs, rs, out_lens, mask_a=True)
        if self.use_cross_modal and cross_y is not None:
            hq, out_s = self.ln3.forward_tm(out, planes="nat", skip=True)
            c, _ = self.cross_attn.forward_tm(hq, out_lens, self.ln3.forward_tm(cross_y, planes="nat"), cross_lens)
            cs, rs = self._dp(self.drop_path_attn, c)
            out = ops.scale_add(out_s, c, cs, rs, out_lens, mask_a=True)
        T2 = out.shape[1]
        tr = self.training                                                              # mlp dropouts: blocks.py:533-540
        h2, out_s = self.ln2.forward_tm(out, planes="nat", skip=True)
        m = ops.linear(h2, self.mlp[0].weight, self.mlp[0].bias, ACT_GELU,
                       drop_p=self.mlp[2].p if tr else 0.0, drop_site="mlp_drop")
        m = ops.linear(m, self.mlp[3].weight, self.mlp[3].bias, ACT_NONE, out_lens, T2,
                       drop_p=self.mlp[4].p if tr else 0.0, drop_site="mlp_drop")
        cs, rs = self._dp(self.drop_path_mlp, m)
        out = ops.scale_add(out_s, m, cs, rs)
        if self.n_ds_strides[0] == 1 and self.n_ds_strides[1] == 1:
            out2 = self.channel_attn.forward_tm(h)
            out = ops.axpby(out, out2, self.t_c_alpha, 1.0 - self.t_c_alpha)
        return out, out_lens

    def forward(self, x, mask, cross_y=None, cross_y_mask=None, pos_embd=None):
        T = x.shape[-1]
        cy = cl = None
        if cross_y is not None:
            cy, cl = to_tm(cross_y), mask_to_lens(cross_y_mask)
        y, out_lens = self.forward_tm(to_tm(x), mask_to_lens(mask), cy, cl)
        out_mask = lens_to_mask(out_lens, T // self.n_ds_strides[0])
        y = from_tm(y)
        if pos_embd is not None:
            y = y + pos_embd * out_mask.to(y.dtype)
        return y, out_mask


class ConvBlock(nn.Module):
    """exported by the reference (blocks.py:1209) but only reachable with backbone_type 'conv',
    which no shipped config uses: not on the accelerated path."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("ConvBlock / backbone_type='conv' is outside the MQ hot path")
