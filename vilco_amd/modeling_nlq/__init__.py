"""Drop-in modules of the NLQ model variant (reference: NLQ/libs/modeling) on the HIP path: the video branch's
sliding-window self-attention (LocalMaskedMHCA), the NLQ TransformerBlock (no channel-attention mix) and the
two-stream ConvTransformerBackbone, behind the same registry API (`register_backbone` / `make_backbone`)."""
from ..modeling.blocks import (AffineDropPath, LayerNorm, MaskedConv1D, MaskedMHA, MaskedMHCA, Scale)  # noqa: F401
from .blocks import LocalMaskedMHCA, TransformerBlock  # noqa: F401
from .models import make_backbone, register_backbone  # noqa: F401
from . import backbones  # noqa: F401

__all__ = ['MaskedConv1D', 'MaskedMHCA', 'MaskedMHA', 'LocalMaskedMHCA', 'LayerNorm', 'TransformerBlock', 'Scale',
           'AffineDropPath', 'make_backbone', 'register_backbone']
