"""EMA of the adapter modules (`post_train_step`, meta_archs.py:702-707).  The reference imports
timm.utils.model_ema.ModelEmaV2 (meta_archs.py:18); timm is not a dependency here, so the same
update rule (ema = decay * ema + (1 - decay) * model over state_dict values) is provided."""
import copy

import torch
from torch import nn


class ModelEmaV2(nn.Module):
    def __init__(self, model, decay=0.9999, device=None):
        super().__init__()
        self.module = copy.deepcopy(model)
        self.module.eval()
        self.decay, self.device = decay, device
        if device is not None:
            self.module.to(device=device)

    @torch.no_grad()
    def _update(self, model, fn):
        for e, m in zip(self.module.state_dict().values(), model.state_dict().values()):
            if self.device is not None:
                m = m.to(device=self.device)
            e.copy_(fn(e, m))

    @torch.no_grad()
    def update(self, model):
        """ema = decay * ema + (1 - decay) * model over every state_dict value: two multi-tensor launches instead of
        three small kernels per tensor"""
        src = model.module if isinstance(model, ModelEmaV2) else model
        es, ms = list(self.module.state_dict().values()), list(src.state_dict().values())
        fl = [(e, m) for e, m in zip(es, ms) if e.is_floating_point() and self.device is None]
        if fl:
            torch._foreach_mul_([e for e, _ in fl], self.decay)
            torch._foreach_add_([e for e, _ in fl], [m for _, m in fl], alpha=1. - self.decay)
        rest = [(e, m) for e, m in zip(es, ms) if not (e.is_floating_point() and self.device is None)]
        for e, m in rest:
            if self.device is not None:
                m = m.to(device=self.device)
            e.copy_(self.decay * e + (1. - self.decay) * m)

    def set(self, model):
        self._update(model, lambda e, m: m)
