"""Temporal point generator (reference: MQ/libs/modeling/loc_generators.py:28-92): per pyramid
level a non-persistent buffer [T_l, 4] = (t, reg_lo, reg_hi, stride), sliced to the level length."""
import torch
from torch import nn

from .models import register_generator


class BufferList(nn.Module):
    def __init__(self, buffers):
        super().__init__()
        for i, b in enumerate(buffers):
            self.register_buffer(str(i), b, persistent=False)

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


@register_generator('point')
class PointGenerator(nn.Module):
    def __init__(self, max_seq_len, fpn_strides, regression_range, use_offset=False, use_us_fpn=False):
        super().__init__()
        assert len(regression_range) == len(fpn_strides)
        assert not use_us_fpn
        self.max_seq_len, self.fpn_levels = max_seq_len, len(fpn_strides)
        self.fpn_strides, self.regression_range, self.use_offset = fpn_strides, regression_range, use_offset
        pts = []
        for stride, rng in zip(fpn_strides, regression_range):
            t = torch.arange(0, max_seq_len, stride)[:, None].to(torch.float32)
            if use_offset:
                t = t + 0.5 * stride
            n = t.shape[0]
            rr = torch.as_tensor(rng, dtype=torch.float)[None].repeat(n, 1)
            st = torch.as_tensor(stride, dtype=torch.float)[None].repeat(n, 1)
            pts.append(torch.cat((t, rr, st), dim=1))
        self.buffer_points = BufferList(pts)

    def forward(self, feats, lengths=None):
        """feats: channel-first tensors (length = shape[-1]) or pass explicit `lengths`."""
        assert len(feats) == self.fpn_levels
        lens = lengths if lengths is not None else [f.shape[-1] for f in feats]
        out = []
        for n, buf in zip(lens, self.buffer_points):
            assert n <= buf.shape[0], "Reached max buffer length for point generator"
            out.append(buf[:n, :])
        return out
