"""Per-iteration LR schedules of the MQ driver (reference: MQ/libs/utils/lr_schedulers.py:10-211), in
closed form.  The reference uses PyTorch's "chainable" recurrences; unrolled they give

  warm-up  (e < W):  lr_e = start + e * (base - start) / (W - 1)          (so lr reaches base at e = W-1)
  cosine   (e >= W): lr_e = eta_min + (base - eta_min) * (1 + cos(pi * (e - W) / (M - W))) / 2
  multistep(e >= W): lr_e = base * gamma ** #{milestones <= e}

tests/test_train_utils.py pins these against sequences produced by the imported reference classes."""
import math
from bisect import bisect_right

from torch.optim.lr_scheduler import LRScheduler


class LinearWarmupCosineAnnealingLR(LRScheduler):
    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=1e-8, last_epoch=-1):
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        super().__init__(optimizer, last_epoch)

    def _at(self, base, e):
        W, M = self.warmup_epochs, self.max_epochs
        if e < W:
            return self.warmup_start_lr + e * (base - self.warmup_start_lr) / max(W - 1, 1)
        return self.eta_min + (base - self.eta_min) * (1 + math.cos(math.pi * (e - W) / (M - W))) / 2

    def get_lr(self):
        return [self._at(b, self.last_epoch) for b in self.base_lrs]

    _get_closed_form_lr = get_lr


class LinearWarmupMultiStepLR(LRScheduler):
    def __init__(self, optimizer, warmup_epochs, milestones, warmup_start_lr=0.0, gamma=0.1, last_epoch=-1):
        self.warmup_epochs, self.warmup_start_lr = warmup_epochs, warmup_start_lr
        self.milestones, self.gamma = sorted(milestones), gamma
        super().__init__(optimizer, last_epoch)

    def _at(self, base, e):
        W = self.warmup_epochs
        if e < W:
            return self.warmup_start_lr + e * (base - self.warmup_start_lr) / max(W - 1, 1)
        return base * self.gamma ** bisect_right(self.milestones, e - W)

    def get_lr(self):
        return [self._at(b, self.last_epoch) for b in self.base_lrs]

    _get_closed_form_lr = get_lr


class WarmupLRScheduler(LRScheduler):
    """linear warm-up to the base rates, constant afterwards (NLQ/libs/utils/lr_schedulers.py:123-185, schedule_type
    "constant" of NLQ's make_scheduler)"""

    def __init__(self, optimizer, warmup_epochs, warmup_start_lr=0.0, last_epoch=-1):
        self.warmup_epochs, self.warmup_start_lr = warmup_epochs, warmup_start_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        e, W = self.last_epoch, self.warmup_epochs
        if e < W:
            return [self.warmup_start_lr + e * (b - self.warmup_start_lr) / max(W - 1, 1) for b in self.base_lrs]
        return list(self.base_lrs)

    _get_closed_form_lr = get_lr
