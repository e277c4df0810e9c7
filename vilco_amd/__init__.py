"""vilco_amd: the MQ video-text transformer hot path of ViLCo (MQ/train_cl.py -> MQ/libs/modeling,
MQ/libs/utils/nms) on MI355X / gfx950.  Hand-written HIP kernels behind a C ABI
(include/vilco_hip.h, vilco_amd/csrc) hosted by PyTorch-ROCm; the python modules keep the
reference's registry names, constructor kwargs, forward contracts and state_dict keys."""
from . import _lib  # noqa: F401
from .ops import get_precision, set_precision  # noqa: F401

__version__ = "0.1"
