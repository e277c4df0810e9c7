"""Drop-in modules of the NLQ model variant (reference: NLQ/libs/modeling) on the HIP path: the video branch's
sliding-window self-attention (LocalMaskedMHCA), the NLQ TransformerBlock (no channel-attention mix) and the
two-stream ConvTransformerBackbone and the LocPointTransformer meta-architecture, behind the same registry API
(`register_backbone` / `make_backbone` / `make_meta_arch`)."""
from ..modeling.blocks import (AffineDropPath, LayerNorm, MaskedConv1D, MaskedMHA, MaskedMHCA, Scale)  # noqa: F401
from .blocks import LocalMaskedMHCA, TransformerBlock  # noqa: F401
from .models import make_backbone, make_meta_arch, register_backbone, register_meta_arch  # noqa: F401
from . import backbones  # noqa: F401
from . import meta_archs  # noqa: F401

__all__ = ['MaskedConv1D', 'MaskedMHCA', 'MaskedMHA', 'LocalMaskedMHCA', 'LayerNorm', 'TransformerBlock', 'Scale',
           'AffineDropPath', 'make_backbone', 'register_backbone', 'make_meta_arch', 'register_meta_arch']
