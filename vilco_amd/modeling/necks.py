"""FPNIdentity neck: one LayerNorm per pyramid level, masks passed through
(reference: MQ/libs/modeling/necks.py:118-198; the only neck the shipped configs can build)."""
from torch import nn

from .blocks import LayerNorm, from_tm, to_tm
from .models import register_neck


@register_neck('identity')
class FPNIdentity(nn.Module):
    def __init__(self, in_channels, out_channel, scale_factor=2.0, start_level=0, end_level=-1,
                 with_ln=True, use_us_fpn=False):
        super().__init__()
        if use_us_fpn:
            raise NotImplementedError("use_us_fpn is off in every shipped config (core/config.py:118)")
        self.in_channels, self.out_channel = in_channels, out_channel
        self.scale_factor, self.use_us_fpn = scale_factor, use_us_fpn
        self.start_level = start_level
        self.end_level = len(in_channels) if end_level == -1 else end_level
        assert self.end_level <= len(in_channels)
        assert 0 <= self.start_level < self.end_level
        self.fpn_norms = nn.ModuleList()
        for i in range(self.start_level, self.end_level):
            assert self.in_channels[i] == self.out_channel
            self.fpn_norms.append(LayerNorm(out_channel) if with_ln else nn.Identity())

    def forward_tm(self, feats, lens):
        assert len(feats) == len(self.in_channels) == len(lens)
        out, out_lens = [], []
        for i, norm in enumerate(self.fpn_norms):
            x = feats[i + self.start_level]
            out.append(norm.forward_tm(x) if isinstance(norm, LayerNorm) else x)
            out_lens.append(lens[i + self.start_level])
        return out, out_lens

    def forward(self, inputs, fpn_masks):
        assert len(inputs) == len(self.in_channels) == len(fpn_masks)
        feats = tuple()
        masks = tuple()
        for i, norm in enumerate(self.fpn_norms):
            x = inputs[i + self.start_level]
            feats += (from_tm(norm.forward_tm(to_tm(x))) if isinstance(norm, LayerNorm) else x,)
            masks += (fpn_masks[i + self.start_level],)
        return feats, masks


@register_neck('fpn')
class FPN1D(nn.Module):
    """the reference's FPN1D cannot be constructed by PtTransformer (TypeError on `use_us_fpn`,
    necks.py:17-25 vs meta_archs.py:547); we raise the same kind of error instead of guessing."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise TypeError("FPN1D.__init__() got an unexpected keyword argument 'use_us_fpn'")
