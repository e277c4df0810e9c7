"""same exports as MQ/libs/modeling/__init__.py:1-12"""
from . import backbones, loc_generators, meta_archs, necks  # noqa: F401  (fill the registries)
from .blocks import (AffineDropPath, ConvBlock, LayerNorm, MaskedConv1D, MaskedMHA, MaskedMHCA,  # noqa: F401
                     Scale, TransformerBlock)
from .meta_archs import BiasLayer  # noqa: F401
from .models import make_backbone, make_generator, make_meta_arch, make_neck  # noqa: F401

__all__ = ['MaskedConv1D', 'MaskedMHCA', 'MaskedMHA', 'LayerNorm', 'TransformerBlock', 'ConvBlock', 'Scale',
           'AffineDropPath', 'make_backbone', 'make_neck', 'make_meta_arch', 'make_generator', 'BiasLayer']
