"""Two-stream ConvTransformerBackbone of the NLQ model (reference: NLQ/libs/modeling/backbones.py:409-615):
video embedding convs + text embedding projections, text stem, video stem with cross-attention to the text, and the
stride-2 branch (cross-attention in its first arch[3] blocks), with per-level local attention windows."""
import torch
from torch import nn

from .. import ops
from ..modeling.blocks import LayerNorm, MaskedConv1D, from_tm, get_sinusoid_encoding, lens_to_mask, mask_to_lens, to_tm
from .blocks import TransformerBlock
from .models import register_backbone


@register_backbone("convTransformer")
class ConvTransformerBackbone(nn.Module):
    def __init__(self, n_vid_in, n_txt_in, n_embd, n_head, n_embd_ks, max_len, arch=(2, 2, 2, 0, 5),
                 mha_win_size=[-1] * 6, scale_factor=2, with_ln=False, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0,
                 use_abs_pe=False, use_rel_pe=False, use_adapter=False):
        super().__init__()
        assert len(arch) == 5
        assert len(mha_win_size) == (1 + arch[3] + arch[4])
        self.arch, self.mha_win_size, self.max_len = arch, mha_win_size, max_len
        self.relu = nn.ReLU(inplace=True)
        self.scale_factor, self.use_abs_pe, self.use_rel_pe = scale_factor, use_abs_pe, use_rel_pe
        if self.use_abs_pe:
            pos_embd = get_sinusoid_encoding(self.max_len, n_embd) / (n_embd ** 0.5)
            self.register_buffer("pos_embd", pos_embd, persistent=False)
            self.register_buffer("pos_embd_tm", pos_embd[0].t().contiguous(), persistent=False)

        def embd(n_in, ks):
            convs, norms = nn.ModuleList(), nn.ModuleList()
            for idx in range(arch[0]):
                convs.append(MaskedConv1D(n_in if idx == 0 else n_embd, n_embd, ks, stride=1, padding=ks // 2,
                                          bias=(not with_ln)))
                norms.append(LayerNorm(n_embd) if with_ln else nn.Identity())
            return convs, norms
        self.vid_embd, self.vid_embd_norm = embd(n_vid_in, n_embd_ks)
        self.txt_embd, self.txt_embd_norm = embd(n_txt_in, 1)

        def block(strides, win, cross):
            return TransformerBlock(n_embd, n_head, n_ds_strides=strides, attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop,
                                    path_pdrop=path_pdrop, mha_win_size=win, use_rel_pe=self.use_rel_pe,
                                    use_cross_modal=cross, use_adapter=use_adapter if (strides == (1, 1) and cross) else False)
        self.use_adapter = use_adapter
        self.vid_stem = nn.ModuleList([block((1, 1), self.mha_win_size[0], True) for _ in range(arch[2])])
        self.txt_stem = nn.ModuleList([block((1, 1), -1, False) for _ in range(arch[1])])
        s = (scale_factor, scale_factor)
        # the reference indexes mha_win_size[1 + idx] with idx restarting at 0 in the second loop (backbones.py:520-545)
        self.branch = nn.ModuleList([block(s, self.mha_win_size[1 + idx], True) for idx in range(arch[3])] +
                                    [block(s, self.mha_win_size[1 + idx], False) for idx in range(arch[4])])
        self.apply(self.__init_weights__)

    def __init_weights__(self, module):
        if isinstance(module, (nn.Linear, nn.Conv1d)) and module.bias is not None:
            torch.nn.init.constant_(module.bias, 0.)

    @staticmethod
    def _conv_ln_relu(conv, norm, x, lens):
        x, lens = conv.forward_tm(x, lens)
        if isinstance(norm, LayerNorm):
            return norm.forward_tm(x, relu=True), lens
        return torch.relu(x), lens

    def forward_tm(self, vid, vid_lens, txt, txt_lens):
        """vid [B,T,Cv], txt [B,L,Ct] token-major + int32 lengths -> lists of pyramid features / lengths"""
        T = vid.shape[1]
        for conv, norm in zip(self.vid_embd, self.vid_embd_norm):
            vid, vid_lens = self._conv_ln_relu(conv, norm, vid, vid_lens)
        if self.use_abs_pe:
            if self.training or T < self.max_len:
                assert T <= self.max_len, "Reached max length."
                pe = self.pos_embd_tm[:T]
            elif T == self.max_len:
                pe = self.pos_embd_tm
            else:
                pe = torch.nn.functional.interpolate(self.pos_embd, T, mode='linear', align_corners=False)[0].t().contiguous()
            vid = ops.add_pe(vid, pe.contiguous(), vid_lens)
        assert txt is not None
        for conv, norm in zip(self.txt_embd, self.txt_embd_norm):
            txt, txt_lens = self._conv_ln_relu(conv, norm, txt, txt_lens)
        for blk in self.txt_stem:
            txt, txt_lens = blk.forward_tm(txt, txt_lens)
        for blk in self.vid_stem:
            vid, vid_lens = blk.forward_tm(vid, vid_lens, txt, txt_lens)
        feats, all_lens = [vid], [vid_lens]
        for blk in self.branch:
            vid, vid_lens = blk.forward_tm(vid, vid_lens, txt, txt_lens)
            feats.append(vid)
            all_lens.append(vid_lens)
        return feats, all_lens

    def forward(self, src_vid, src_vid_mask, src_txt, src_txt_mask):
        feats, all_lens = self.forward_tm(to_tm(src_vid), mask_to_lens(src_vid_mask), to_tm(src_txt), mask_to_lens(src_txt_mask))
        return (tuple(from_tm(f) for f in feats), tuple(lens_to_mask(l, f.shape[1]) for f, l in zip(feats, all_lens)))
