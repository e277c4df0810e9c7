from .nms import batched_nms, nms_1d_cpu  # noqa: F401
