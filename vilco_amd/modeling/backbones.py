"""ConvTransformerBackbone of the MQ model on the HIP path (reference: MQ/libs/modeling/backbones.py:12-289).
Same constructor kwargs (meta_archs.py:467-487), module names and state_dict keys.  Inside, tensors
are token-major [B,T,C] with int32 valid lengths; `forward` keeps the reference's channel-first
signature and returns channel-first pyramids + bool masks."""
import os

import torch
from torch import nn

from .. import ops
from .blocks import (LayerNorm, MaskedConv1D, TransformerBlock, from_tm, get_sinusoid_encoding,
                     lens_to_mask, mask_to_lens, to_tm)
from .modeling_xlnet_x import XLNetConfig, XLNetModel
from .models import register_backbone


# off by default: measured ~0.8 ms faster per step in most runs, but one run in four lands 3-7 ms SLOWER (r02, same box:
# 32.3 / 39.3 / 32.0 ms) -- the interleaving of the two queues is not under our control.  VILCO_TEXT_STREAM=1 enables it.
_TEXT_STREAM = os.environ.get("VILCO_TEXT_STREAM", "0") == "1"
_side_streams = {}


def _text_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def _find_xlnet_config(n_embd):
    """the reference opens 'configs/xlnet_config_<D>.json' relative to CWD (backbones.py:132);
    VILCO_XLNET_CONFIG_DIR may point elsewhere.  None when no file exists for this width."""
    name = 'xlnet_config_%s.json' % n_embd
    for d in (os.environ.get("VILCO_XLNET_CONFIG_DIR"), 'configs'):
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return None


@register_backbone("convTransformer")
class ConvTransformerBackbone(nn.Module):
    def __init__(self, n_in, n_embd, n_head, n_embd_ks, max_len, use_xl, arch=(2, 2, 5), t_c_alpha=0.8,
                 scale_factor=2, with_ln=False, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0,
                 use_abs_pe=False, use_rel_pe=False, use_dcn=False, dcn_start_layer=0,
                 use_cross_modal=False, n_txt_in=768, xlnet_config=None):
        super().__init__()
        assert len(arch) == 3
        if isinstance(n_in, (list, tuple)) or use_dcn:
            raise NotImplementedError("multi-stream inputs / deformable convs are not on the MQ hot path")
        self.t_c_alpha, self.arch, self.max_len = t_c_alpha, arch, max_len
        self.relu = nn.ReLU(inplace=True)
        self.scale_factor = scale_factor
        self.use_abs_pe, self.use_rel_pe = use_abs_pe, use_rel_pe
        self.use_xl, self.use_cross_modal = use_xl, use_cross_modal
        self.n_in, self.proj = n_in, None

        if self.use_abs_pe:
            pos_embd = get_sinusoid_encoding(self.max_len, n_embd) / (n_embd ** 0.5)
            self.register_buffer("pos_embd", pos_embd, persistent=False)
            # token-major copy for the HIP add_pe kernel
            self.register_buffer("pos_embd_tm", pos_embd[0].t().contiguous(), persistent=False)

        self.embd, self.embd_norm = nn.ModuleList(), nn.ModuleList()
        for idx in range(arch[0]):
            self.embd.append(MaskedConv1D(n_in if idx == 0 else n_embd, n_embd, n_embd_ks, stride=1,
                                          padding=n_embd_ks // 2, bias=(not with_ln)))
            self.embd_norm.append(LayerNorm(n_embd) if with_ln else nn.Identity())

        def block(strides, cross):
            return TransformerBlock(n_embd, n_head, n_ds_strides=strides, attn_pdrop=attn_pdrop,
                                    proj_pdrop=proj_pdrop, path_pdrop=path_pdrop, t_c_alpha=t_c_alpha,
                                    use_rel_pe=self.use_rel_pe, use_cross_modal=cross)
        self.stem = nn.ModuleList([block((1, 1), self.use_cross_modal) for _ in range(arch[1])])
        self.branch = nn.ModuleList([block((scale_factor, scale_factor), self.use_cross_modal)
                                     for _ in range(arch[2])])

        if not self.use_xl and len(self.stem) > 0:
            # stem[0] runs twice in this configuration (forward_tm below, backbones.py:276-278): the gradients through its
            # mask-ignoring channel attention span ~1e10 between the first padded row and the rest (blocks.ChannelAttention)
            self.stem[0].channel_attn.attn.wide_range = True
        if self.use_xl:
            if xlnet_config is None:
                path = _find_xlnet_config(n_embd)
                if path is None:
                    raise FileNotFoundError("configs/xlnet_config_%s.json not found (backbones.py:132 "
                                            "resolves it relative to CWD)" % n_embd)
                xlnet_config = XLNetConfig.from_json_file(path)
            elif isinstance(xlnet_config, dict):
                xlnet_config = XLNetConfig.from_dict(xlnet_config)
            self.xlnet = XLNetModel(xlnet_config)

        if self.use_cross_modal:
            self.txt_embd, self.txt_embd_norm = nn.ModuleList(), nn.ModuleList()
            for idx in range(arch[0]):
                self.txt_embd.append(MaskedConv1D(n_txt_in if idx == 0 else n_embd, n_embd, 1, stride=1,
                                                  padding=0, bias=(not with_ln)))
                self.txt_embd_norm.append(LayerNorm(n_embd) if with_ln else nn.Identity())
            # txt_stem blocks are built with the reference's default t_c_alpha (backbones.py:161-170)
            self.txt_stem = nn.ModuleList([
                TransformerBlock(n_embd, n_head, n_ds_strides=(1, 1), attn_pdrop=attn_pdrop,
                                 proj_pdrop=proj_pdrop, path_pdrop=path_pdrop, use_rel_pe=self.use_rel_pe,
                                 use_cross_modal=False) for _ in range(arch[1])])
        self.apply(self.__init_weights__)

    def __init_weights__(self, module):
        if isinstance(module, (nn.Linear, nn.Conv1d)) and module.bias is not None:
            torch.nn.init.constant_(module.bias, 0.)

    @staticmethod
    def _next_planes(convs, i):
        """the operand-plane layout the conv after convs[i] reads its input in (hint for ops.layernorm), or None"""
        if i + 1 >= len(convs):
            return None
        c = convs[i + 1]
        if c.conv.groups != 1 or c.stride != 1:
            return None
        return {1: "nat", 3: "seq"}.get(c.conv.kernel_size[0])

    @staticmethod
    def _conv_ln_relu(conv, norm, x, lens, planes=None):
        """planes: the layout of the output's operand planes when it goes straight into another conv (ops.layernorm)"""
        x, lens = conv.forward_tm(x, lens)
        if isinstance(norm, LayerNorm):
            return norm.forward_tm(x, relu=True, planes=planes), lens
        return torch.relu(x), lens

    def forward_tm(self, x, lens, text=None, text_lens=None):
        """x [B,T,Cin] token-major, lens int32 [B]; text [B,L,Ctxt] -> lists of feats_tm / lens per level."""
        B, T, _ = x.shape
        for i, (conv, norm) in enumerate(zip(self.embd, self.embd_norm)):
            x, lens = self._conv_ln_relu(conv, norm, x, lens, self._next_planes(self.embd, i))

        if self.use_abs_pe:
            if self.training or T < self.max_len:
                assert T <= self.max_len, "Reached max length."
                pe = self.pos_embd_tm[:T]
            elif T == self.max_len:
                pe = self.pos_embd_tm      # F.interpolate to the same length is the identity
            else:
                pe = torch.nn.functional.interpolate(self.pos_embd, T, mode='linear', align_corners=False)
                pe = pe[0].t().contiguous()
            x = ops.add_pe(x, pe.contiguous(), lens)

        # The text side (77 tokens per clip: ~150 launch-latency-bound kernels forward, as many backward) depends on
        # nothing the video embedding and stem produce, and is first read by branch 0's cross-attention: it runs on a
        # second HIP stream underneath the video stem's large kernels (autograd replays every node's backward on the
        # stream of its forward, so the backward overlaps the same way).  Optional, see _TEXT_STREAM.
        q = q_lens = None
        side = None
        if self.use_cross_modal and text is not None:
            main = torch.cuda.current_stream()
            side = _text_stream(x.device) if (x.is_cuda and (_TEXT_STREAM or ops.fork_enabled("text"))) else None
            if side is not None:
                side.wait_stream(main)
            with torch.cuda.stream(side if side is not None else main):
                q, q_lens = text, text_lens
                for i, (conv, norm) in enumerate(zip(self.txt_embd, self.txt_embd_norm)):
                    q, q_lens = self._conv_ln_relu(conv, norm, q, q_lens, self._next_planes(self.txt_embd, i))
                for blk in self.txt_stem:
                    q, q_lens = blk.forward_tm(q, q_lens)

        x = ops.seg_cut(x, next_stage=True)              # (staged backward, see below: embeddings | stem blocks | branch blocks)
        for blk in self.stem[:-1]:
            x, lens = blk.forward_tm(x, lens)            # stem blocks are called without cross_y
            x = ops.seg_cut(x, next_stage=True)
        for blk in self.stem[-1:]:
            x, lens = blk.forward_tm(x, lens)

        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            q.record_stream(torch.cuda.current_stream())

        # ops.seg_cut: identity, or (a GraphedStep capturing the backward in stages) the cut between two stages: the text
        # features and every pyramid level go on as leaves -- stage 0 = embeddings + text side + stem, then one per branch block
        if q is not None:
            q = ops.seg_cut(q)
        x = ops.seg_cut(x, next_stage=True)
        feats, all_lens = [x], [lens]
        for idx, blk in enumerate(self.branch):
            if idx == 0:
                if self.use_xl:
                    x = self.xlnet.forward_tm(x, lens)
                else:
                    x, lens = self.stem[0].forward_tm(x, lens)   # backbones.py:276-278: stem[0] re-applied
            if idx in (1, 2):                                # backbones.py:280-283: no cross-attn on 1, 2
                x, lens = blk.forward_tm(x, lens)
            else:
                x, lens = blk.forward_tm(x, lens, q, q_lens)
            x = ops.seg_cut(x, next_stage=True)
            feats.append(x)
            all_lens.append(lens)
        return feats, all_lens

    def forward(self, x, mask, src_text=None, src_text_mask=None, use_xl=False):
        text = text_lens = None
        if src_text is not None:
            text, text_lens = to_tm(src_text), mask_to_lens(src_text_mask)
        feats, all_lens = self.forward_tm(to_tm(x), mask_to_lens(mask), text, text_lens)
        out_feats = tuple(from_tm(f) for f in feats)
        out_masks = tuple(lens_to_mask(l, f.shape[1]) for f, l in zip(feats, all_lens))
        return out_feats, out_masks
