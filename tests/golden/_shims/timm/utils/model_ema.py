"""Import shim (test infrastructure only): minimal ModelEmaV2 with timm 1.0.3 semantics
(deepcopy, eval, ema = decay*ema + (1-decay)*model over state_dict values)."""
import copy
import torch
from torch import nn


class ModelEmaV2(nn.Module):
    def __init__(self, model, decay=0.9999, device=None):
        super().__init__()
        self.module = copy.deepcopy(model)
        self.module.eval()
        self.decay = decay
        self.device = device
        if device is not None:
            self.module.to(device=device)

    def _update(self, model, update_fn):
        with torch.no_grad():
            for e, m in zip(self.module.state_dict().values(), model.state_dict().values()):
                if self.device is not None:
                    m = m.to(device=self.device)
                e.copy_(update_fn(e, m))

    def update(self, model):
        self._update(model, update_fn=lambda e, m: self.decay * e + (1.0 - self.decay) * m)

    def set(self, model):
        self._update(model, update_fn=lambda e, m: m)
