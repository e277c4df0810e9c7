"""A training iteration captured as hipGraphs and replayed -- the host side of the step without the host.

The reference's step (MQ/libs/utils/train_utils.py:322-357: zero_grad, forward, backward, clip_grad_norm_, optimizer
step) is ~1500 kernel launches on this path, each enqueued by a few microseconds of Python + ctypes: ~25 ms of host time
per iteration, about what the GPU needs for config P and 4x what it needs for the T = 256 configuration.  The reference has
no counterpart of this file (eager PyTorch); this is the MI355X answer to its step loop.

`GraphedStep(model, optimizer)` runs the first iterations of every input signature eagerly, then captures

  graph 1: [bump the dropout step word] -> layout change -> backbone -> heads -> fused labels + losses -> backward
  graph 2: global-norm clip coefficient -> fused AdamW / SGD update (+ max|w| partials for the next weight packs)

and from then on an iteration is: copy the batch into the static input buffers, replay 1, (eager work the caller wants
between backward and update: a gradient all-reduce, an EWC / MAS penalty), hand the learning rates over, replay 2.
What must differ between replays lives in device memory: the inputs, the dropout step word (common.h: vilco_step_seed),
the stochastic-depth factors (torch's graph-safe Philox draw, captured), the loss-normaliser EMA, the optimizer's step
counts and learning rates (vilco_optim_step_dev).

What is captured is what ran: graphs are keyed by the input shapes, task id and the identity / requires_grad of every
parameter, and dropped when parameters were written from outside (load_state_dict, an eager optimizer step) -- cached
operand planes and max|w| partials would be stale otherwise.  Steps the device half cannot run alone (distillation
against host-side logits, narration SSL, BiC) fall back to the eager path.
"""
import ctypes as C
import gc
import os

import torch

from . import _lib, ops


_DP_REPLAY_SYNC = os.environ.get("VILCO_DP_REPLAY_SYNC", "1") != "0"
_OWN_STREAM = os.environ.get("VILCO_GRAPH_OWN_STREAM", "1") != "0"      # replays leave the default (null) stream: GraphedStep._replay
# data-parallel replays: the backward captured as one hipGraph per stage (heads + losses, then block by block -- ops.seg_cut),
# the gradient buckets a stage completes launched right behind its replay, under the stages that follow.  0: one graph, the
# whole exchange after it.
# Default for replays with a live reducer (VILCO_DP_SEGMENTS=0: one graph, the whole exchange behind it).  Bit-exact against the
# eager reducer on the models of tests/test_dist_gpu.py; at config P equal to the one-graph replay up to the order in which the
# cut leaves accumulate their gradients (<= 2.4e-4 of a tensor's maximum, tools/lab/dp_staged_dbg2.py, 60 calls, run-to-run
# identical) -- ON A STREAM OF ITS OWN: replayed on the default (null) stream with the buckets queued asynchronously between
# the stage graphs, the small gradients (the ones gathered into their bucket by a copy) came back stale or as garbage once the
# host ran ahead (not RCCL, not the multi-tensor copy, not Python's GC; exact on any created stream).  _replay therefore moves
# every replay off the null stream.  VILCO_DP_STAGE_SYNC=1 adds a host wait before every bucket group (the first
# workaround; not needed any more).
_DP_SEGMENTS = os.environ.get("VILCO_DP_SEGMENTS", "1") != "0"


_replay_streams = {}
_SHARED_STREAM = os.environ.get("VILCO_GRAPH_SHARED_STREAM", "1") != "0"


def _replay_stream(device):
    """The stream GraphedStep replays on: ONE per device for every instance of the process (VILCO_GRAPH_SHARED_STREAM=0: one per
    instance, round 5's form).  Round 6: with a stream per instance, the first hipGraphLaunch on the stream of the 5th-6th instance of
    a process that had also initialised and destroyed an RCCL group segfaulted inside the runtime (tests/test_dist_gpu.py followed by
    tests/test_graph_gpu.py, deterministic; exact on the null stream and with one shared stream) -- replays are serial on the host
    anyway, so nothing is lost by sharing."""
    if not _SHARED_STREAM:
        return torch.cuda.Stream(device=device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _replay_streams.get(key)
    if st is None:
        st = _replay_streams[key] = torch.cuda.Stream(device=device)
    return st


def _capture_kw():
    """with a process group alive, RCCL's watchdog thread polls the events of finished collectives; under the default
    ("global") capture error mode such a query from ANOTHER thread while this one captures kills the capture -- and the
    watchdog (seen on a one-rank group: 'operation not permitted when stream is capturing').  thread_local: only this
    thread's calls are policed."""
    import torch.distributed as dist
    return {"capture_error_mode": "thread_local"} if (dist.is_available() and dist.is_initialized()) else {}


class GraphedStep:
    def __init__(self, model, optimizer=None, clip_grad_l2norm=-1.0, eager_steps=2, between=None, enabled=True,
                 gt_pad=8, max_graphs=16, reducer=None, comm_in_graph=None, segments=None):
        """optimizer: a FusedOptimizer (None: forward + backward only, gradients left in p.grad);
        eager_steps: iterations of a new signature run eagerly before its capture (>= 1 with an optimizer: the capture
        must follow an eager update, whose max|w| partials scale the captured weight packs);
        between: callable run eagerly between backward and the update of every iteration;
        gt_pad: ground-truth rows are padded to this many segments per clip so that their count does not key the graph;
        reducer: a dist.GradReducer -- eager iterations exchange gradients from its autograd hooks (overlapped with
        backward), replayed ones stage by stage (`segments`, below) or right after graph 1 (`reduce_now`, averaged in place);
        comm_in_graph (default: env VILCO_DP_GRAPH_COMM, off): capture the bucketed RCCL all-reduces INSIDE graph 1 -- the
        reducer's autograd hooks stay live during the capture, every collective lands on RCCL's stream behind an event of
        the capture stream, and a replay overlaps the exchange with the rest of backward the way the eager step does.  Needs
        the "nccl" backend; falls back to the exchange after the replay when the capture refuses.  Exercised on one rank
        (tests/test_dist_gpu.py); not yet on a multi-GPU node, hence opt-in.
        segments (default: env VILCO_DP_SEGMENTS, on; only with an enabled reducer and without comm_in_graph): the backward is
        captured in stages -- graph 1 = forward + heads / losses backward, then one graph per backbone stage, cut at the
        pyramid levels (ops.seg_cut) -- and a replayed iteration launches every gradient bucket as soon as the stage that
        completes it has been enqueued: the all-reduces run on the collective's stream under the remaining stages, no
        collective is captured, and the numbers are those of the single-graph step (same kernels, same order per stage)."""
        self.model, self.optimizer, self.clip = model, optimizer, float(clip_grad_l2norm)
        self.eager_steps = max(int(eager_steps), 1 if optimizer is not None else 0)
        self.between, self.enabled, self.gt_pad, self.max_graphs = between, bool(enabled), gt_pad, int(max_graphs)
        self.reducer = reducer
        self.comm_in_graph = (os.environ.get("VILCO_DP_GRAPH_COMM", "0") == "1") if comm_in_graph is None else bool(comm_in_graph)
        self.segments = _DP_SEGMENTS if segments is None else bool(segments)
        self._graphs = {}
        self._pool = None
        self._lr_dev = None
        self.params = [p for p in model.parameters()]
        self.stats = dict(eager=0, captured=0, replayed=0, dropped=0)

    # ------------------------------------------------------------------ bookkeeping
    def _param_sig(self):
        ps = [p for p in self.model.parameters()]
        if len(ps) != len(self.params) or any(a is not b for a, b in zip(ps, self.params)):
            self.params = ps
        return hash(tuple((p.data_ptr(), p.requires_grad) for p in ps))

    @staticmethod
    def _version_sum(tracked):
        return sum(p._version for p in tracked)

    def reset(self):
        """forget every captured graph; their memory pool goes with the last of them"""
        self.stats['dropped'] += sum(1 for e in self._graphs.values() if 'graph' in e)
        self._graphs = {}
        self._pool = None

    # ------------------------------------------------------------------ one iteration
    def __call__(self, video_list, task_id=0, prev_out_cls_logits=None):
        model = self.model
        inp = model.prepare(video_list, True, gt_pad=self.gt_pad)
        if not (self.enabled and model.training and model.capturable(inp, task_id, prev_out_cls_logits)):
            return self._eager(inp, video_list, task_id, prev_out_cls_logits)
        key = (inp.signature(), int(task_id), self._param_sig(), int(model.n_known > 0), ops.arithmetic_key())
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = {'seen': 0}
        if 'graph' in ent and not self._still_valid(ent):
            self.reset()
            ent = self._graphs[key] = {'seen': 0}
        if 'graph' not in ent and self.reducer is not None and self.reducer.enabled and self.reducer.planned() is None:
            ent['seen'] += 1
            return self._eager(inp, video_list, task_id, prev_out_cls_logits)      # no bucket plan yet: it is built by an eager finish()
        if 'graph' not in ent:
            # a signature is captured once it has come back often enough, and only while there is room: batches whose
            # shapes keep changing (ragged text lengths) stay eager instead of evicting the graphs of the common shapes
            full = sum(1 for e in self._graphs.values() if 'graph' in e) >= self.max_graphs
            if ent['seen'] < self.eager_steps or full:
                ent['seen'] += 1
                return self._eager(inp, video_list, task_id, prev_out_cls_logits)
            self._capture(ent, inp, task_id)
        return self._replay(ent, inp)

    def try_capture(self, video_list, task_id=0):
        """Capture the graphs of this batch's signature NOW, without replaying them (no collective is issued: a staged capture
        holds none), and report whether that worked instead of raising.  For callers that must agree ACROSS RANKS on replay vs
        eager before the first replayed step (bench.py, N > 1): a capture that fails on one rank only would leave the ranks
        issuing different collectives.  False also when the step is not capturable; the signature then stays eager."""
        model = self.model
        inp = model.prepare(video_list, True, gt_pad=self.gt_pad)
        if not (self.enabled and model.training and model.capturable(inp, task_id, None)):
            return False
        if self.reducer is not None and self.reducer.enabled and self.reducer.planned() is None:
            return False                         # (the bucket plan comes from an eager finish(): run an eager step first)
        key = (inp.signature(), int(task_id), self._param_sig(), int(model.n_known > 0), ops.arithmetic_key())
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = {'seen': 0}
        if 'graph' in ent and self._still_valid(ent):
            return True
        try:
            self._capture(ent, inp, task_id)
            return True
        except Exception as e:                   # noqa: BLE001 -- reported, not raised: the caller falls back to eager steps
            import warnings
            warnings.warn("GraphedStep.try_capture failed (%s: %s): this signature stays eager" % (type(e).__name__, e))
            self._graphs.pop(key, None)
            torch.cuda.synchronize()
            for p in self.params:
                p.grad = None
            return False

    def _eager(self, inp, video_list, task_id, prev):
        self.stats['eager'] += 1
        for p in self.params:
            p.grad = None
        if self.reducer is not None:
            self.reducer.begin()
        losses = self.model.forward_prepared(inp, video_list, task_id=task_id, prev_out_cls_logits=prev)
        losses['final_loss'].backward()
        if self.reducer is not None:
            self.reducer.finish()
        if self.between is not None:
            self.between()
        if self.optimizer is not None:
            self.optimizer.step(clip_grad_l2norm=self.clip)
        # detached: a caller holding the loss dict must not keep this iteration's autograd graph (and with it the
        # parameters' AccumulateGrad nodes, which remember the stream they were made on) alive into a later capture
        return {k: v.detach() for k, v in losses.items()}

    def _still_valid(self, ent):
        if ent['versions'] != self._version_sum(ent['tracked']):
            return False                     # a parameter was written through torch (load_state_dict, an in-place edit)
        if self.optimizer is None:
            return ent['wgen'] == ops._weight_gen[0]      # weights written through raw pointers since the capture: the
                                                          # captured forward reads operand planes packed before it
        return ent['plans'] is self.optimizer._plans      # (a captured update re-packs its weights inside graph 1)

    # ------------------------------------------------------------------ capture
    def _capture(self, ent, inp, task_id):
        from .modeling import blocks
        from .modeling.meta_archs import StepInputs
        model, lib = self.model, _lib.load()
        static = StepInputs()
        static.T, static.narr = inp.T, None
        for name in StepInputs.__slots__:
            if name not in ("T", "narr"):
                t = getattr(inp, name)
                setattr(static, name, None if t is None else t.clone())
        for p in self.params:
            p.grad = None
        model._cat = None                    # last iteration's head tensors (and through them its autograd graph)
        blocks.reset_drop_pool()
        gc.collect()
        torch.cuda.synchronize()
        if ops.range_check in ("warn", "auto"):
            import warnings
            warnings.warn("GraphedStep: VILCO_RANGE_CHECK=%s is a debug mode of the EAGER step (it reads gradients on the host); a "
                          "captured step skips the check and runs every backward product in the default format" % ops.range_check)
        red = self.reducer if (self.reducer is not None and self.reducer.enabled) else None
        # at most two attempts: with the collectives inside the capture, then (this runtime refused) without them
        while True:
            g = torch.cuda.CUDAGraph()
            comm = bool(red is not None and self.comm_in_graph and red._avg and not ent.get('comm_refused'))
            if self.reducer is not None:
                # default: no collectives inside the capture (replays exchange after graph 1), but the captured weight-gradient
                # kernels write into the reducer's bucket slots: a replayed backward leaves the large gradients in place.
                # comm: the hooks stay live and the all-reduces are captured with the backward they overlap.
                self.reducer.begin(hooks=comm)
            failed = None
            staged = bool(red is not None and self.segments and not comm)
            tape = ops.SegTape() if staged else None
            seg_graphs, seg_done = [], []
            forks = ops._FORKS
            pins = ops._capture_pins = []    # persistent buffers the recorded kernels address: they live as long as the graph
            try:
                ops.seg_tape = tape
                if staged and os.environ.get("VILCO_DP_STAGE_FORKS", "1") == "0":
                    ops._FORKS = set()           # (debugging aid: a staged capture without the forked text / regression-head chains)
                with torch.cuda.graph(g, pool=self._pool, **_capture_kw()):
                    if red is not None:
                        red._capture_stream = torch.cuda.current_stream()
                    _lib.check(lib.vilco_seed_word_bump(ops._stream()))
                    losses = model.forward_prepared(static, None, task_id=task_id)
                    ops.seg_tape = None          # (the cuts are on the tape; nothing after the forward makes new ones)
                    losses['final_loss'].backward()      # staged: stops at the leaves of the last cut
                    if comm:
                        red.finish()             # waits become edges of the graph; p.grad = views of the averaged buckets
                    ops.join_side_streams()      # forked chains (ops.fork_enabled) end here
                    keys = sorted(losses)
                    out = torch.stack([losses[k].detach().reshape(()).float() for k in keys])
                if staged:
                    # one graph per stage, last stage first: its roots are the tensors the stage handed on, seeded with what
                    # the later stages accumulated in the leaves that replaced them
                    seg_done.append(self._grad_marks())
                    for st in sorted({r[2] for r in tape.records}, reverse=True):
                        recs = [r for r in tape.records if r[2] == st and r[1].grad is not None]
                        if not recs:
                            continue
                        gs = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gs, pool=self._pool if self._pool is not None else g.pool(), **_capture_kw()):
                            red._capture_stream = torch.cuda.current_stream()
                            torch.autograd.backward([r[0] for r in recs], [r[1].grad for r in recs])
                            ops.join_side_streams()
                        seg_graphs.append(gs)
                        seg_done.append(self._grad_marks())
            except Exception as e:               # noqa: BLE001 -- re-raised below unless it is the collectives' capture
                failed = e
            finally:
                ops.seg_tape = None
                ops._capture_pins = None
                ops._FORKS = forks
                if self.reducer is not None:
                    self.reducer._capture_stream = None
                    self.reducer.end_capture()
            if failed is None:
                break
            if not comm:
                raise failed
            ent['comm_refused'] = True           # this runtime does not capture the collectives: exchange after the replay
            torch.cuda.synchronize()
            for p in self.params:
                p.grad = None
        ent['comm'] = comm
        ent['pins'] = pins
        ent['seg_graphs'] = seg_graphs if staged else None
        ent['seg_marks'] = seg_done if staged else None
        del losses, tape                 # the cut leaves' .grad buffers stay where the stage graphs read and write them (pool)
        blocks.reset_drop_pool()
        if self._pool is None:
            self._pool = g.pool()
        ent['fill'] = []
        if self.reducer is not None and self.reducer.enabled:
            # The exchange covers the reducer's plan (the union over ranks of the gradient-bearing parameters), the captured
            # update covers what has a gradient HERE.  The two must be the same set on every rank or replicas drift apart:
            # a planned parameter this rank's step does not reach gets a zero gradient of its own (re-zeroed before every
            # replay; the exchange writes the other ranks' average into it), exactly what the eager finish() leaves in
            # p.grad; a gradient outside the plan cannot be averaged at all.
            planned = {id(p) for p in self.reducer.planned()}
            stray = [i for i, p in enumerate(self.params) if p.grad is not None and id(p) not in planned]
            if stray:
                raise RuntimeError("GraphedStep: %d parameter(s) received a gradient but are not in the GradReducer's plan "
                                   "(the set of trained parameters changed: call reducer.rebuild())" % len(stray))
            for p in self.params:
                if p.grad is None and id(p) in planned:
                    p.grad = torch.zeros_like(p)
                    ent['fill'].append(p.grad)
        ent.update(graph=g, static=static, out=out, keys=keys, grads=[p.grad for p in self.params])
        if ent.get('seg_graphs') is not None:
            ent['seg_upto'] = self._bucket_schedule(ent['seg_marks'])
        if self.optimizer is not None:
            opt = self.optimizer
            opt.prepare_step(pin=True)       # plans + pointer tables for THESE gradient tensors, built outside the capture
            if self._lr_dev is None:
                self._lr_dev = torch.zeros(16, dtype=torch.float32, device=self.params[0].device)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=self._pool, **_capture_kw()):
                opt.step(clip_grad_l2norm=self.clip, lr_dev=self._lr_dev)
            for pl in opt._plans:            # the capture counted an update that did not run
                pl['count'] -= 1
            ent.update(opt_graph=g2, plans=opt._plans)
        # the parameters the captured step reads: those that got a gradient, and frozen ones (the adapters' EMA copies are
        # neither: they are written in place after every iteration and only read at inference)
        ent['tracked'] = [p for p, g in zip(self.params, ent['grads']) if g is not None or not p.requires_grad]
        ent['versions'] = self._version_sum(ent['tracked'])
        ent['wgen'] = ops._weight_gen[0]
        self.stats['captured'] += 1

    def _grad_marks(self):
        """(address, version) of every parameter's gradient right now: a stage touched the gradients whose mark moved"""
        return [None if p.grad is None else (p.grad.data_ptr(), p.grad._version) for p in self.params]

    def _bucket_schedule(self, marks):
        """marks[k] = _grad_marks() after stage k of the captured backward (0 = graph 1).  -> upto[k]: the buckets [0, upto[k])
        of the reducer's plan are complete once stage k has run (plan order = launch order on every rank; a bucket waits
        for the stage that last touches any of its parameters, and for every bucket before it)."""
        last = {}
        prev = [None] * len(self.params)
        for k, m in enumerate(marks):
            for p, a, b in zip(self.params, prev, m):
                if b is not None and a != b:
                    last[id(p)] = k
            prev = m
        ready = [max([last.get(id(p), 0) for p in b["params"]] or [0]) for b in self.reducer.buckets]
        upto, n = [], 0
        for k in range(len(marks)):
            while n < len(ready) and max(ready[:n + 1]) <= k:
                n += 1
            upto.append(n)
        upto[-1] = len(ready)
        return upto

    # ------------------------------------------------------------------ replay
    def _replay_staged(self, ent, inp):
        """graph 1, then stage by stage; behind every stage the buckets it completed (collectives enqueued on the current
        stream's order by torch.distributed: they start when that stage is done and run under the next ones)"""
        red = self.reducer
        for p, g in zip(self.params, ent['grads']):
            if p.grad is not g:
                p.grad = g
        if ent['fill']:
            torch._foreach_zero_(ent['fill'])
        red.reduce_begin()
        graphs = [ent['graph']] + list(ent['seg_graphs'])
        if (not red._avg and _DP_REPLAY_SYNC) or os.environ.get("VILCO_DP_STAGE_LATE") == "1":      # (_LATE: debugging aid)
            # gloo (CPU-side collectives: tests, two replicas on one GPU): a gloo collective next to a running graph crawls
            # (seconds per step, see _replay) -- the stages are replayed back to back and the exchange follows, unoverlapped
            for gk in graphs:
                gk.replay()
            torch.cuda.current_stream().synchronize()
            red.reduce_launch(len(red.buckets))
        else:
            mode = os.environ.get("VILCO_DP_STAGE_SYNC", "0")      # see _DP_SEGMENTS: "1" a host wait before every bucket group,
            hsync = mode == "1"                                    # "step" one per step (on the previous step's end), "0" none (default)
            if mode == "step":
                ev = getattr(self, "_step_done", None)
                if ev is not None:
                    ev.synchronize()
            for k, gk in enumerate(graphs):
                gk.replay()
                if ent['seg_upto'][k] > red._next:
                    if hsync:
                        torch.cuda.current_stream().synchronize()
                    red.reduce_launch(ent['seg_upto'][k])
        red.reduce_wait()
        if os.environ.get("VILCO_DP_STAGE_SYNC", "0") == "step":
            self._step_done = torch.cuda.Event()
            self._step_done.record()

    def _replay(self, ent, inp):
        """Never on the default (null) stream: graph launches with asynchronous eager work queued between them -- bucket gathers
        and copy-backs, a `between` penalty adding into the gradients, the learning-rate store -- went wrong there once the host
        ran ahead of the device (exact on any created stream).  Called with the null stream current, the replay
        moves to a stream of its own, ordered behind and in front of the caller's by events."""
        cur = torch.cuda.current_stream()
        own = cur
        # (round 6: the bare forward + backward replay leaves the null stream too -- it was verified exact there, but a created
        # stream is also 0.1-0.2 ms per step faster, tools/lab/stream_ab.py, and no replay depends on null-stream ordering)
        if cur == torch.cuda.default_stream(cur.device) and _OWN_STREAM:
            if getattr(self, "_own_stream", None) is None:
                self._own_stream = _replay_stream(cur.device)
            own = self._own_stream
            own.wait_stream(cur)
        with torch.cuda.stream(own):
            self._replay_body(ent, inp)
        if own is not cur:
            cur.wait_stream(own)
        self.stats['replayed'] += 1
        out = ent['out'].clone()
        return {k: out[i] for i, k in enumerate(ent['keys'])}

    def _replay_body(self, ent, inp):
        static = ent['static']
        for name, t in inp.tensors():
            getattr(static, name).copy_(t, non_blocking=True)
        staged = ent.get('seg_graphs') is not None and self.reducer is not None and self.reducer.enabled
        if staged:
            self._replay_staged(ent, inp)
        else:
            ent['graph'].replay()
            for gk in (ent.get('seg_graphs') or ()):      # captured in stages, the exchange switched off since: plain replays
                gk.replay()
        for p, g in zip(self.params, ent['grads']):
            if p.grad is not g:
                p.grad = g
        if self.reducer is not None and self.reducer.enabled and not ent.get('comm') and not staged:
            if ent['fill']:
                torch._foreach_zero_(ent['fill'])
            # The exchange reads what the replay writes, so it cannot start earlier anyway -- and a collective queued behind a
            # graph that is still running has been seen to crawl: two replicas on one GPU over gloo, 6-10 s per step without
            # this wait, 0.24 s with it (tools/lab/dp_replay_probe.py).  VILCO_DP_REPLAY_SYNC=0: enqueue behind the graph.
            if _DP_REPLAY_SYNC:
                torch.cuda.current_stream().synchronize()
            self.reducer.reduce_now()
        if self.between is not None:
            self.between()
        if self.optimizer is not None:
            opt = self.optimizer
            ng = len(opt.param_groups)
            lr = (C.c_float * ng)(*[float(g['lr']) for g in opt.param_groups])
            _lib.check(_lib.load().vilco_store_f32(self._lr_dev.data_ptr(), lr, ng, ops._stream()))
            ent['opt_graph'].replay()
            opt.note_replays(1)
            opt._opt_called = True           # torch's LR schedulers check that optimizer.step() ran before scheduler.step()
