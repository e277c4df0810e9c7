from .nms import batched_nms, nms_1d_cpu  # noqa: F401
from .train_utils import (fix_random_seed, make_optimizer, make_scheduler, save_checkpoint,  # noqa: F401
                          train_step)
