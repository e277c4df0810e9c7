"""NLQ operators on the HIP path (reference: NLQ/libs/modeling/blocks.py -- LocalMaskedMHCA :417-755,
TransformerBlock :757-875).  State_dict keys and constructor arguments as the reference."""
from torch import nn

from .. import ops
from ..modeling.blocks import (AffineDropPath, LayerNorm, MaskedMHA, MaskedMHCA, _drop_rowscale, from_tm, lens_to_mask,
                               mask_to_lens, to_tm)
from ..ops import ACT_GELU, ACT_NONE


class LocalMaskedMHCA(MaskedMHCA):
    """MaskedMHCA whose softmax runs over a sliding window of `window_size` keys centred on the query
    (blocks.py:417-755; the Longformer chunking of :511-690 is an implementation device of the reference -- the
    arithmetic is a banded attention, which the flash kernels evaluate directly: mask mode 4 visits only the key tiles
    the window reaches, and nothing of size T x window is ever materialised)."""

    def __init__(self, n_embd, n_head, window_size, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0,
                 use_rel_pe=False):
        super().__init__(n_embd, n_head, n_qx_stride=n_qx_stride, n_kv_stride=n_kv_stride, attn_pdrop=attn_pdrop,
                         proj_pdrop=proj_pdrop)
        assert window_size > 1 and n_head >= 1
        if use_rel_pe:
            raise NotImplementedError("use_rel_pe is off in every shipped NLQ config")
        self.window_size, self.window_overlap, self.use_rel_pe = window_size, window_size // 2, use_rel_pe

    def _attend(self, q, k, v, q_lens, kv_lens):
        assert q.shape[1] % (2 * self.window_overlap) == 0, "sequence length must be a multiple of 2 * window_overlap (blocks.py:594)"
        q = ops.linear(q, self.query.weight, self.query.bias)
        k = ops.linear(k, self.key.weight, self.key.bias)
        v = ops.linear(v, self.value.weight, self.value.bias)
        o = ops.attention(q, k, v, kv_lens, self.n_head, self.scale, mode=ops.MASK_LOCAL, window=self.window_overlap,
                          drop_p=self.attn_drop.p if self.training else 0.0)
        out = ops.linear(o, self.proj.weight, self.proj.bias, ACT_NONE, q_lens, q.shape[1],
                         drop_p=self.proj_drop.p if self.training else 0.0, drop_site="proj_drop")
        return out, q_lens


class TransformerBlock(nn.Module):
    """pre-LN block: (local) conv-attention, optional text cross-attention, MLP (blocks.py:757-875)."""

    def __init__(self, n_embd, n_head, n_ds_strides=(1, 1), n_out=None, n_hidden=None, act_layer=nn.GELU, attn_pdrop=0.0,
                 proj_pdrop=0.0, path_pdrop=0.0, mha_win_size=-1, use_rel_pe=False, use_cross_modal=False,
                 use_adapter=False):
        super().__init__()
        assert len(n_ds_strides) == 2
        if use_adapter:
            raise NotImplementedError("NLQ adapter modules are outside the accelerated path")
        self.ln1, self.ln2 = LayerNorm(n_embd), LayerNorm(n_embd)
        if mha_win_size > 1:
            self.attn = LocalMaskedMHCA(n_embd, n_head, window_size=mha_win_size, n_qx_stride=n_ds_strides[0],
                                        n_kv_stride=n_ds_strides[1], attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop,
                                        use_rel_pe=use_rel_pe)
        else:
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
            self.pool_skip = nn.MaxPool1d(n_ds_strides[0] + 1, stride=n_ds_strides[0], padding=(n_ds_strides[0] + 1) // 2)
        else:
            self.pool_skip = nn.Identity()
        n_hidden = 4 * n_embd if n_hidden is None else n_hidden
        n_out = n_embd if n_out is None else n_out
        self.mlp = nn.Sequential(nn.Conv1d(n_embd, n_hidden, 1), act_layer(), nn.Dropout(proj_pdrop, inplace=True),
                                 nn.Conv1d(n_hidden, n_out, 1), nn.Dropout(proj_pdrop, inplace=True))
        if path_pdrop > 0.0:
            self.drop_path_attn = AffineDropPath(n_embd, drop_prob=path_pdrop)
            self.drop_path_mlp = AffineDropPath(n_out, drop_prob=path_pdrop)
        else:
            self.drop_path_attn, self.drop_path_mlp = nn.Identity(), nn.Identity()
        self.use_adapter = use_adapter

    def _dp(self, mod, x):
        if isinstance(mod, AffineDropPath):
            return mod.scale, _drop_rowscale(x, mod.drop_prob, self.training)
        return None, None

    def forward_tm(self, x, lens, cross_y=None, cross_lens=None):
        # the skip connections take their input back from the LayerNorm that opens the branch (ops.layernorm `skip`: the
        # gradient over the skip is added inside that LayerNorm's backward kernel)
        if self.attn.fusable() and self.ln1.affine:
            a, out_lens, _, xs = self.attn.forward_tm_fused(x, lens, self.ln1, False)
        else:
            h, xs = self.ln1.forward_tm(x, skip=True)
            a, out_lens = self.attn.forward_tm(h, lens)
        skip = ops.maxpool3s2(xs, lens) if self.n_ds_strides[0] > 1 else xs
        cs, rs = self._dp(self.drop_path_attn, a)
        out = ops.scale_add(skip, a, cs, rs, out_lens, mask_a=True)
        if self.use_cross_modal and cross_y is not None:
            hq, out_s = self.ln3.forward_tm(out, skip=True)
            c, _ = self.cross_attn.forward_tm(hq, out_lens, self.ln3.forward_tm(cross_y), cross_lens)
            cs, rs = self._dp(self.drop_path_attn, c)
            out = ops.scale_add(out_s, c, cs, rs, out_lens, mask_a=True)
        tr = self.training
        h2, out_s = self.ln2.forward_tm(out, skip=True)
        m = ops.linear(h2, self.mlp[0].weight, self.mlp[0].bias, ACT_GELU,
                       drop_p=self.mlp[2].p if tr else 0.0, drop_site="mlp_drop")
        m = ops.linear(m, self.mlp[3].weight, self.mlp[3].bias, ACT_NONE, out_lens, out.shape[1],
                       drop_p=self.mlp[4].p if tr else 0.0, drop_site="mlp_drop")
        cs, rs = self._dp(self.drop_path_mlp, m)
        return ops.scale_add(out_s, m, cs, rs), out_lens

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
