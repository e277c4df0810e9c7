"""Data-parallel gradient exchange for the MQ training step.

One process per GPU; each rank runs fwd+bwd on its own clips; gradients of the parameters that
actually receive one are averaged with bucketed all-reduces over RCCL (torch.distributed backend
"nccl" on ROCm), launched from autograd hooks as soon as a bucket's gradients are complete so the
collectives overlap the rest of backward.  The reference MQ driver has no working DDP (DDP(model) is
commented out, train_utils.py:298); its intent -- torchrun + DistributedSampler (datasets.py:24) -- and
VQ's `find_unused_parameters=True` DDP (VQ/train_cl.py:112-117) are what this mirrors: 107 of the 465
tensors never get a gradient at arch (2,2,5), so the bucket plan is pruned statically after the first
backward instead of searching the graph every step.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, bucket_mb=64, group=None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        self.buckets = None          # built after the first backward (static pruning of unused params)
        self._hooks = []
        self._pending = []

    # -- plan -------------------------------------------------------------------------------
    def _build(self):
        used = [p for p in self.model.parameters() if p.requires_grad and p.grad is not None]
        used.reverse()               # parameters() order is roughly forward order -> reverse ~ backward order
        self.buckets, cur, cur_bytes = [], [], 0
        for p in used:
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= self.bucket_bytes:
                self.buckets.append(self._make_bucket(cur))
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(self._make_bucket(cur))
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    @staticmethod
    def _make_bucket(params):
        n = sum(p.numel() for p in params)
        flat = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return {"params": list(params), "flat": flat, "views": views, "left": len(params), "work": None}

    def _make_hook(self, bi):
        def hook(param):
            b = self.buckets[bi]
            b["left"] -= 1
            if b["left"] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        torch._foreach_copy_(b["views"], [p.grad for p in b["params"]])
        b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(b)

    # -- per step ---------------------------------------------------------------------------
    def begin(self):
        if self.buckets is not None:
            for b in self.buckets:
                b["left"], b["work"] = len(b["params"]), None
            self._pending = []

    def finish(self):
        """wait for the collectives and leave the averaged gradient in every p.grad"""
        if self.buckets is None:
            self._build()
            for b in self.buckets:       # first step: nothing was launched from hooks yet
                self._launch(b)
        else:
            for b in self.buckets:       # a bucket whose hook count did not reach 0 would deadlock peers
                assert b["work"] is not None, "a parameter that had a gradient in step 1 got none now"
        for b in self._pending:
            b["work"].wait()
            b["flat"].div_(self.world)
            torch._foreach_copy_([p.grad for p in b["params"]], b["views"])
        self._pending = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
