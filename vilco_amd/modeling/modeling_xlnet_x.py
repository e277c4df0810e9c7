"""The one XLNet layer the MQ backbone runs at full T before branch 0 (`use_xl`, backbones.py:130-135,
267-273), restated for the HIP path.  Reference: MQ/libs/modeling/modeling_xlnet_x.py --
XLNetModel.forward :1075-1308 with attn_type "bi", no mems, no segments, no target mapping,
XLNetRelativeAttention :210-467 (rel_shift_bnij :256-268, rel_attn_core :270-320),
XLNetFeedForward :470-490.  Parameter names/shapes match the reference state_dict
(`xlnet.layer.0.rel_attn.q [D,H,hd]`, ...), including the never-used word_embedding / mask_emb /
r_s_bias / seg_embed tensors, so checkpoints load unchanged.
"""
import json

import torch
from torch import nn

from .. import ops
from ..ops import ACT_GELU
import os as _os

_ATEN_BIAS = _os.environ.get("VILCO_XL_BIAS_ATEN") == "1"      # q + bias through ATen (the stale-sum bug, ops._BiasAdd)


class XLNetConfig:
    """the fields of configs/xlnet_config_<D>.json that the layer reads"""

    def __init__(self, d_model, n_head, d_head=None, d_inner=None, n_layer=1, dropout=0.1,
                 layer_norm_eps=1e-12, vocab_size=32000, initializer_range=0.02, attn_type="bi",
                 bi_data=False, clamp_len=-1, ff_activation="gelu", **unused):
        self.d_model, self.n_head = d_model, n_head
        self.d_head = d_head if d_head is not None else d_model // n_head
        self.d_inner = d_inner if d_inner is not None else 4 * d_model
        self.n_layer, self.dropout, self.layer_norm_eps = n_layer, dropout, layer_norm_eps
        self.vocab_size, self.initializer_range = vocab_size, initializer_range
        self.attn_type, self.bi_data, self.clamp_len = attn_type, bi_data, clamp_len
        self.ff_activation = ff_activation
        if attn_type != "bi" or bi_data or clamp_len > 0 or ff_activation != "gelu":
            raise NotImplementedError("only the shipped XLNet setting (bi, no bi_data, no clamp, gelu)")

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls.from_dict(json.load(f))


class XLNetRelativeAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        D, H, hd = config.d_model, config.n_head, config.d_head
        if D % H != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)" % (D, H))
        self.n_head, self.d_head, self.d_model = H, hd, D
        self.scale = 1 / (hd ** 0.5)
        std = config.initializer_range
        for name, shape in (("q", (D, H, hd)), ("k", (D, H, hd)), ("v", (D, H, hd)), ("o", (D, H, hd)),
                            ("r", (D, H, hd)), ("r_r_bias", (H, hd)), ("r_s_bias", (H, hd)),
                            ("r_w_bias", (H, hd)), ("seg_embed", (2, H, hd))):
            setattr(self, name, nn.Parameter(torch.empty(*shape).normal_(mean=0.0, std=std)))
        self.layer_norm = nn.LayerNorm(D, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.dropout)

    def forward_tm(self, h, pos_emb, lens):
        """h [B,T,D] token-major, pos_emb [2T, D], lens int32 [B] -> LayerNorm(attn_out + h)."""
        q = ops.linear_kn(h, self.q)                         # ONE projection; the reference's two streams differ by a bias
        # (ops.bias_add, not `q + bias`: the bias gradient must not be an ATen multi-block `sum` inside a captured step, ops._BiasAdd)
        if _ATEN_BIAS:                                       # (rounds 1-5, kept as the demonstration of the stale-sum bug)
            q_w, q_r = q + self.r_w_bias.view(1, 1, -1), q + self.r_r_bias.view(1, 1, -1)
        else:
            q_w = ops.bias_add(q, self.r_w_bias)             # q + r_w_bias (content stream, modeling_xlnet_x.py:284)
            q_r = ops.bias_add(q, self.r_r_bias)             # q + r_r_bias (position stream, :287)
        k = ops.linear_kn(h, self.k)
        v = ops.linear_kn(h, self.v)
        k_r = ops.linear_kn(pos_emb, self.r)                 # [2T, H*hd] ([B, 2T, H*hd] under dropout)
        p = self.dropout.p if self.training else 0.0
        vec = ops.rel_attention(q_w, q_r, k, v, k_r, lens, self.n_head, self.scale, drop_p=p)   # attn_prob dropout (:308)
        C = self.n_head * self.d_head
        out = ops.linear(vec, self.o.view(self.d_model, C), drop_p=p, drop_site="xl_attn_out")   # einsum("ibnd,hnd->ibh") + dropout (:327)
        out = ops.axpby(out, h, 1.0, 1.0)
        return ops.layernorm(out, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, planes="nat")   # -> layer_1


class XLNetFeedForward(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer_norm = nn.LayerNorm(config.d_model, eps=config.layer_norm_eps)
        self.layer_1 = nn.Linear(config.d_model, config.d_inner)
        self.layer_2 = nn.Linear(config.d_inner, config.d_model)
        self.dropout = nn.Dropout(config.dropout)
        for lin in (self.layer_1, self.layer_2):
            lin.weight.data.normal_(mean=0.0, std=config.initializer_range)
            lin.bias.data.zero_()

    def forward_tm(self, x):
        p = self.dropout.p if self.training else 0.0
        y = ops.linear(x, self.layer_1.weight, self.layer_1.bias, ACT_GELU, drop_p=p, drop_site="xl_ff_inner")   # (:486)
        y = ops.linear(y, self.layer_2.weight, self.layer_2.bias, drop_p=p, drop_site="xl_ff_out")              # (:488)
        y = ops.axpby(y, x, 1.0, 1.0)
        return ops.layernorm(y, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, planes="nat")   # -> the next layer's q / k / v


class XLNetLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.rel_attn = XLNetRelativeAttention(config)
        self.ff = XLNetFeedForward(config)
        self.dropout = nn.Dropout(config.dropout)

    def forward_tm(self, h, pos_emb, lens):
        return self.ff.forward_tm(self.rel_attn.forward_tm(h, pos_emb, lens))


class XLNetModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.d_model, self.n_layer = config.d_model, config.n_layer
        self.word_embedding = nn.Embedding(config.vocab_size, config.d_model)
        self.word_embedding.weight.data.normal_(mean=0.0, std=config.initializer_range)
        self.mask_emb = nn.Parameter(torch.empty(1, 1, config.d_model).normal_(0.0, config.initializer_range))
        self.layer = nn.ModuleList([XLNetLayer(config) for _ in range(config.n_layer)])
        self.dropout = nn.Dropout(config.dropout)
        self._pos_cache = {}

    def relative_positional_encoding(self, qlen, device):
        """pos_emb[p] = [sin(s_p f), cos(s_p f)], s = arange(qlen, -qlen, -1)  (:1029-1066, bi, no mems)."""
        key = (qlen, str(device))
        if key not in self._pos_cache:
            freq_seq = torch.arange(0, self.d_model, 2.0, dtype=torch.float)
            inv_freq = 1 / torch.pow(10000, (freq_seq / self.d_model))
            pos_seq = torch.arange(qlen, -qlen, -1.0, dtype=torch.float)
            sinusoid = torch.einsum("i,d->id", pos_seq, inv_freq)
            pe = torch.cat([torch.sin(sinusoid), torch.cos(sinusoid)], dim=-1)
            self._pos_cache = {key: pe.to(device).contiguous()}
        return self._pos_cache[key]

    def forward_tm(self, x, lens):
        """inputs_embeds [B,T,D] + attention lengths -> last hidden state [B,T,D]."""
        pos_emb = self.relative_positional_encoding(x.shape[1], x.device)
        p, tr = self.dropout.p, self.training
        h = ops.dropout(x, p, tr, "xl_input")                                                     # (:1201)
        if tr and p > 0.0:
            # the reference drops out the position embedding AFTER expanding it over the batch (:1228): per-clip masks
            pos_emb = ops.dropout(pos_emb.unsqueeze(0).expand(x.shape[0], -1, -1).contiguous(), p, tr, "xl_pos_emb")
        for layer in self.layer:
            h = layer.forward_tm(h, pos_emb, lens)
        return ops.dropout(h, p, tr, "xl_output")                                                 # (:1280)

    def forward(self, inputs_embeds=None, attention_mask=None, **unused):
        lens = attention_mask.to(torch.int32).sum(dim=1, dtype=torch.int32)
        return (self.forward_tm(inputs_embeds.contiguous(), lens),)
