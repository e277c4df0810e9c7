"""Data-parallel gradient exchange for the MQ training step.

One process per GPU; each rank runs fwd+bwd on its own clips; gradients of the parameters that
actually receive one are averaged with bucketed all-reduces over RCCL (torch.distributed backend
"nccl" on ROCm), launched from autograd hooks as soon as a bucket's gradients are complete so the
collectives overlap the rest of backward.  The reference MQ driver has no working DDP (DDP(model) is
commented out, train_utils.py:298); its intent -- torchrun + DistributedSampler (datasets.py:24) -- and
VQ's `find_unused_parameters=True` DDP (VQ/train_cl.py:112-117) are what this mirrors: 107 of the 465
tensors never get a gradient at arch (2,2,5), so the bucket plan is pruned statically after the first
backward instead of searching the graph every step.

Memory traffic: the weight-gradient products -- where the bytes are -- write straight INTO their slot of the
bucket's flat buffer (`begin()` installs `grad_slot` as `ops.grad_slot_provider`; ops._grad_out), so the
collective finds its input in place; the small gradients (biases, LayerNorm, scales) and any gradient autograd
had to sum are copied into their slots by one multi-tensor copy per bucket.  After the all-reduce `p.grad` is a
view of the slot (no copy back) and the average is taken by the collective itself (ReduceOp.AVG on RCCL; gloo
has no AVG, so the CPU test path divides).

Why bucketed all-reduce and not a hand-rolled reduce-scatter + all-gather: on the 8-GPU xGMI mesh every pair of
GPUs has its own link, so the bandwidth-optimal exchange is direct -- each rank sends 1/8 of a bucket to each
peer (reduce-scatter), sums, and sends its eighth back (all-gather): 2 x 7/8 of the bucket over 7 links in
parallel, ~1.4 ms for the 843 MB of config P at ~150 GB/s per link, against ~9.6 ms for a one-link ring
(SURVEY.md 5).  RCCL picks its algorithm per message size from the topology it detects; on a fully connected
mesh it builds as many rings / trees as there are links, which reaches the same link-parallel bandwidth for
64 MB buckets without our issuing 56 point-to-point transfers per bucket from Python.  `profile_buckets()`
reports what a node actually delivers per bucket (bench.py prints it for N > 1), which is the number that
decides whether a direct RS + AG is worth writing; this container has one GPU, so it has not been measured.

Robustness (every rank must issue the same collectives in the same order):
  * the plan is built from the UNION over ranks of the gradient-bearing parameters (one MAX all-reduce of a
    bitmask), so rank-local differences cannot produce different bucket layouts;
  * a planned parameter that got no gradient in some later step is zero-filled and its bucket still
    launched (in `finish`), instead of one rank asserting while its peers wait in a collective;
  * `rebuild()` drops the plan (call it when the set of trained parameters changes, e.g. a new task).
"""
import torch
import torch.distributed as dist

DEFAULT_BUCKET_MB = 64      # xGMI rings are per-link bound: few large collectives beat many small ones
import os as _os
_DEBUG_NO_COLLECTIVE = _os.environ.get("VILCO_DP_DEBUG_NO_COLLECTIVE") in ("1", "2")      # one-rank debugging aid (tools/lab/dp_staged_dbg2.py)
_DEBUG_EMULATE = _os.environ.get("VILCO_DP_DEBUG_NO_COLLECTIVE") == "2"


_DEBUG_SLOW_COPY = _os.environ.get("VILCO_DP_DEBUG_SLOW_COPY") == "1"


def _copy_list(dst, src):
    if _DEBUG_SLOW_COPY:
        for d, s_ in zip(dst, src):
            d.copy_(s_)
    else:
        torch._foreach_copy_(dst, src)


class _NoWork:
    def wait(self):
        return True


class _EmulatedWork:
    """debugging aid (VILCO_DP_DEBUG_NO_COLLECTIVE=2, one rank): the stream choreography of an asynchronous collective without the
    collective -- an event of the caller's stream, a high-priority side stream waiting for it, an end event of the side stream the
    caller waits for in wait() -- to tell torch's streams / events apart from RCCL in the null-stream hazard"""
    _side = None

    def __init__(self):
        if _EmulatedWork._side is None:
            _EmulatedWork._side = torch.cuda.Stream(priority=-1)
        cur = torch.cuda.current_stream()
        e = torch.cuda.Event()
        e.record(cur)
        _EmulatedWork._side.wait_event(e)
        self.end = torch.cuda.Event()
        self.end.record(_EmulatedWork._side)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.end)
        return True


class GradReducer:
    def __init__(self, model, bucket_mb=DEFAULT_BUCKET_MB, group=None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        self.buckets = None          # built after the first backward (static pruning of unused params)
        self.enabled = True          # False: hooks do nothing (rank-local steps, e.g. profiling on rank 0 only)
        self._avg = dist.get_backend(group) == "nccl"
        self._hooks = []
        self._pending = []
        self._slot = {}              # id(param) -> (bucket index, view)
        self._slot_by_ptr = {}       # param.data_ptr() -> (bucket index, view)
        self._handed = set()
        self._hooks_live = True
        self._next = 0               # index of the next bucket to launch (strict order)
        self._order, self._order_hooks = [], []     # first backward: the order in which the gradients became complete

    # -- plan -------------------------------------------------------------------------------
    def _build(self):
        cand = [p for p in self.model.parameters() if p.requires_grad]
        flags = torch.tensor([1 if p.grad is not None else 0 for p in cand], dtype=torch.int32,
                             device=cand[0].device)
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)      # union over ranks
        used = [p for p, f in zip(cand, flags.tolist()) if f]
        used.reverse()               # parameters() order is roughly forward order -> reverse ~ backward order
        # Better: the order in which the first backward completed the gradients (recorded by begin()'s hooks; rank 0's order
        # for everybody).  Buckets then fill in the order backward produces them -- which is what lets a hook-driven or a
        # stage-by-stage replayed backward (vilco_amd/graph.py) launch them early; module registration order does not (the
        # text stem and the neck come last in parameters() and finish with the LAST stage of backward).
        for h in self._order_hooks:
            h.remove()
        self._order_hooks = []
        pos = {pid: i for i, pid in enumerate(self._order)}
        rank_of = torch.tensor([pos.get(id(p), 1 << 30) for p in cand], dtype=torch.int64, device=cand[0].device)
        if self.world > 1:
            dist.broadcast(rank_of, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        rk = dict(zip((id(p) for p in cand), rank_of.tolist()))
        if any(v < (1 << 30) for v in rk.values()):
            used.sort(key=lambda p: rk[id(p)])       # stable: parameters the first backward did not reach keep the old order, last
        self._order = []
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
            for p, v in zip(b["params"], b["views"]):
                self._slot[id(p)] = (bi, v)
                self._slot_by_ptr[p.data_ptr()] = (bi, v)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    @staticmethod
    def _make_bucket(params):
        n = sum(p.numel() for p in params)
        flat = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return {"params": list(params), "flat": flat, "views": views, "left": len(params), "work": None,
                "done": set()}

    def grad_slot(self, p):
        """the bucket slot of parameter `p` for a kernel to write its gradient into, or None (no plan yet, `p` not
        planned, reducer disabled, or the slot was already handed out in this step: a parameter used twice gets its
        second gradient in ordinary memory and autograd sums the two)."""
        if not self.enabled or self.buckets is None:
            return None
        key = p.data_ptr()           # (inside autograd the weight arrives as a saved-tensor alias: same memory, another object)
        s = self._slot_by_ptr.get(key)
        if s is None or key in self._handed or s[1].numel() != p.numel():
            return None
        self._handed.add(key)
        return s[1].view(p.shape)

    def _make_hook(self, bi):
        def hook(param):
            if not self.enabled or not self._hooks_live:
                return
            b = self.buckets[bi]
            if id(param) in b["done"]:
                return
            b["done"].add(id(param))
            b["left"] -= 1
            # collectives must be issued in the same order on every rank: bucket i goes out only after 0..i-1 (a
            # bucket that is not complete on this rank holds the later ones back until `finish`)
            while self._next < len(self.buckets) and self.buckets[self._next]["left"] == 0:
                self._launch(self.buckets[self._next])
                self._next += 1
        return hook

    def _sync_producers(self):
        """inside a stream capture the backward runs on several streams (ops.fork_enabled: text branch, regression head) and a
        bucket's last hook fires on whichever of them finished it: the stream that is about to read the whole bucket waits
        for the others (edges of the captured graph; eager backward runs on one stream)"""
        if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            return
        from . import ops
        cur = torch.cuda.current_stream()
        others = [st for (name, dev), st in ops._side_streams.items() if dev == cur.device_index]
        if getattr(self, "_capture_stream", None) is not None:
            others.append(self._capture_stream)
        for st in others:
            if st != cur:
                with torch.cuda.stream(st):
                    live = torch.cuda.is_current_stream_capturing()
                if live:
                    cur.wait_stream(st)

    def _launch(self, b):
        self._sync_producers()
        src, dst = [], []
        for p, v in zip(b["params"], b["views"]):
            if p.grad is None:
                v.zero_()                                   # planned but unused in this step on this rank
            elif p.grad.data_ptr() != v.data_ptr():         # not produced in place
                src.append(p.grad)
                dst.append(v)
        if src:
            _copy_list(dst, src)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if _DEBUG_NO_COLLECTIVE and self.world == 1:
            b["work"] = _EmulatedWork() if _DEBUG_EMULATE else _NoWork()
        else:
            b["work"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)
        self._pending.append(b)

    # -- per step ---------------------------------------------------------------------------
    def begin(self, hooks=True):
        """start of a step's backward.  hooks=False (a hipGraph capture, vilco_amd/graph.py): the dW kernels still write
        into the bucket slots -- a replayed backward then leaves the large gradients where the collective reads them --
        but no collective is issued from the autograd hooks; `end_capture()` undoes it."""
        from . import ops
        self._hooks_live = bool(hooks)
        # live hooks launch a bucket's all-reduce as soon as its last gradient arrives: every gradient must be complete when
        # autograd hands it over, so the deferred finishing of ops.py (_Deferring) is off for such a backward
        ops.defer_blocked = bool(hooks) and self.enabled
        ops.grad_slot_provider = self.grad_slot if self.enabled else None      # dW kernels write into the bucket slots
        self._handed = set()
        self._next = 0
        if self.buckets is not None:
            for b in self.buckets:
                b["left"], b["work"], b["done"] = len(b["params"]), None, set()
            self._pending = []
        elif self.enabled and not self._order_hooks:
            self._order = []
            for p in self.model.parameters():
                if p.requires_grad:
                    self._order_hooks.append(p.register_post_accumulate_grad_hook(lambda param: self._order.append(id(param))))

    def finish(self):
        """wait for the collectives and leave the averaged gradient in every planned p.grad (views of the flat
        buckets: nothing is copied back)"""
        if not self.enabled:
            self._release_slots()
            return
        if self.buckets is None:
            self._build()
        for b in self.buckets[self._next:]:      # first step, or buckets held back by an incomplete one
            self._launch(b)
        self._next = len(self.buckets)
        with torch.no_grad():
            for b in self._pending:
                b["work"].wait()
                if not self._avg:
                    b["flat"].div_(self.world)
                for p, v in zip(b["params"], b["views"]):
                    p.grad = v
        self._pending = []
        self._release_slots()

    def _release_slots(self):
        """no kernel outside a begin() ... finish() / reduce_now() bracket may write into the buckets (a backward run for
        another purpose, e.g. the EWC / MAS importance pass, would otherwise put its dW into the flat buffers)"""
        from . import ops
        if getattr(ops.grad_slot_provider, "__self__", None) is self:
            ops.grad_slot_provider = None
        ops.defer_blocked = False
        self._hooks_live = True

    def end_capture(self):
        self._release_slots()

    def planned(self):
        """the parameters of the bucket plan (None before the first finish())"""
        return None if self.buckets is None else [p for b in self.buckets for p in b["params"]]

    def reduce_now(self):
        """Average the gradients that are already complete in p.grad -- a step replayed as a hipGraph (vilco_amd/graph.py)
        runs no autograd hooks -- and write the result back INTO those gradient tensors: a captured optimizer step holds
        their addresses.  Same buckets, same order as the hook path; costs the copy back (no overlap with backward: a
        replayed backward is one graph launch)."""
        if not self.enabled:
            return
        self.reduce_begin()
        self.reduce_launch(len(self.buckets))
        self.reduce_wait()

    # the same exchange in pieces, for a backward replayed in stages (GraphedStep segments): buckets go out in plan order as soon
    # as the stage that completes their last gradient has been enqueued, and run under the stages that follow
    def reduce_begin(self):
        if self.buckets is None:
            self._build()
        self.begin(hooks=False)

    def reduce_launch(self, upto):
        """launch buckets [already launched, upto) -- every rank with the same `upto` sequence"""
        while self._next < min(int(upto), len(self.buckets)):
            self._launch(self.buckets[self._next])
            self._next += 1

    def reduce_wait(self):
        self.reduce_launch(len(self.buckets))
        with torch.no_grad():
            for b in self._pending:
                b["work"].wait()
                if not self._avg:
                    b["flat"].div_(self.world)
                src, dst = [], []
                for p, v in zip(b["params"], b["views"]):
                    if p.grad is None:
                        p.grad = v.clone()
                    elif p.grad.data_ptr() != v.data_ptr():
                        src.append(v)
                        dst.append(p.grad)
                if src:
                    _copy_list(dst, src)
        self._pending = []
        self._release_slots()

    def profile_buckets(self, iters=3):
        """Every rank calls this (it issues collectives): each bucket's all-reduce alone, fenced by device syncs ->
        [{"mb", "ms", "algbw_GBps", "busbw_GBps"}] (busbw = algbw * 2 (n - 1) / n, the ring convention).  For reading a
        scaling run: the step's exchange cost is the sum of these minus what overlapped with backward."""
        import time
        if self.buckets is None:
            return []
        out = []
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        for b in self.buckets:
            flat = b["flat"]
            sync = (lambda: torch.cuda.synchronize(flat.device)) if flat.is_cuda else (lambda: None)
            dist.all_reduce(flat, op=op, group=self.group)                 # warm the communicator for this size
            sync()
            t0 = time.perf_counter()
            for _ in range(iters):
                dist.all_reduce(flat, op=op, group=self.group)
            sync()
            ms = (time.perf_counter() - t0) / iters * 1e3
            mb = flat.numel() * flat.element_size() / 2 ** 20
            alg = flat.numel() * flat.element_size() / (ms * 1e-3) / 1e9
            out.append({"mb": mb, "ms": ms, "algbw_GBps": alg, "busbw_GBps": alg * 2 * (self.world - 1) / max(self.world, 1)})
        return out

    def rebuild(self):
        """forget the plan (the set of trained parameters changed); the next `finish` builds a new one"""
        self.remove()
        self.buckets, self._slot, self._slot_by_ptr, self._pending = None, {}, {}, []
        self._order = []

    def remove(self):
        from . import ops
        if ops.grad_slot_provider is not None and getattr(ops.grad_slot_provider, "__self__", None) is self:
            ops.grad_slot_provider = None
        for h in self._hooks + self._order_hooks:
            h.remove()
        self._hooks, self._order_hooks = [], []
