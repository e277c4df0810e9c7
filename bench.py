#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: clips/sec (fwd+bwd) of the MQ video-text transformer at
T=2304, C=2304 (config "P" of SURVEY.md section 8: D=1024, H=16, arch (2,2,5), XLNet layer on, text
L=77 x 768, 22 classes, B=2 clips per GPU), synthetic inputs resident in HBM, random-init weights.

  python bench.py --gpus 1 --steps K --warmup W            (single process)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU,
      data parallel over clips, gradient all-reduce over RCCL inside the timed step)

A "step" = forward + backward of one batch through vilco_amd (HIP kernels via libvilco_hip.so);
for N > 1 it includes the bucketed gradient all-reduce.  Rank 0 prints ONE JSON line.  Extra objects:
"roofline" for the dominant kernel (the MFMA GEMM), "cpu_baseline" = the oracle (CPU restatement of
the reference, kind "port") timed on the host cores on a bounded sample, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GFLOP_PER_CLIP_FWD_BWD = 1917.0      # BASELINE.md section 2, config P (measured with FlopCounterMode)
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)

# MQ/configs/xlnet_config_1024.json (dropout 0.1 included)
P_XLNET = dict(d_model=1024, n_head=16, d_head=64, d_inner=2048, n_layer=1, dropout=0.1,
               layer_norm_eps=1e-12, vocab_size=32000, initializer_range=0.02, attn_type="bi",
               bi_data=False, clamp_len=-1, ff_activation="gelu")


def p_xlnet(dropout=0.1):
    return dict(P_XLNET, dropout=dropout)


def p_config(dropout=0.1, droppath=0.1):
    """config P with the reference's training regularisation (mq_vilco.yaml:56-57); the property tests pass 0, 0"""
    from vilco_amd.core.config import make_config
    over = dict(dataset=dict(input_dim=2304, num_classes=22, max_seq_len=2304),
                model=dict(embd_dim=1024, fpn_dim=1024, head_dim=1024, n_head=16, backbone_arch=(2, 2, 5),
                           use_abs_pe=True, use_cross_modal=True, n_txt_in=768, max_buffer_len_factor=1.0,
                           use_xl=True),
                train_cfg=dict(init_loss_norm=100, dropout=dropout, droppath=droppath))
    return make_config(**over)['model']


def synth_batch(B, device, seed=0, T=2304, Cin=2304, L=77):
    """SURVEY.md 8d synthetic clips: feats ~ N(0,1) [Cin, t_b], t_0 = T, t_b = T-17; text [768, 77]."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for b in range(B):
        t = T if b % 2 == 0 else T - 17
        out.append({'video_id': 'v%d' % b, 'feats': torch.randn(Cin, t, generator=g).to(device),
                    'segments': torch.tensor([[10.0, 40.0], [60.5, 130.25]]), 'labels': torch.tensor([1, 5]),
                    'fps': 30.0, 'duration': 100.0, 'feat_stride': 16, 'feat_num_frames': 16,
                    'segmentation_labels': torch.zeros(t, 22),
                    'prompt_feature': torch.randn(768, L, generator=g).to(device)})
    return out


CPU_THREADS_CAP = 64        # torch's CPU ops at these sizes stop scaling (and then slow down) beyond ~one socket's cores


def cpu_baseline_worker():
    """child process: the oracle (CPU restatement of the reference's pure-PyTorch path) on the SAME workload as the
    GPU leg -- config P, 2 clips, train-mode arithmetic (Bernoulli dropout 0.1 / droppath 0.1 / XLNet dropout 0.1),
    fp32, 1 warm-up + up to 3 timed fwd+bwd steps.  Prints one JSON line after EVERY step, so the parent
    can stop it when its time budget is spent and still report what was measured."""
    import vilco_amd.modeling as vm
    from oracle import mq_oracle
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except Exception:
        pass
    physical = logical
    try:      # distinct (physical id, core id) pairs = physical cores
        cores, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for l in f:
                if l.startswith("physical id"):
                    phys = l.split(":")[1].strip()
                elif l.startswith("core id"):
                    core = l.split(":")[1].strip()
                    cores.add((phys, core))
        if cores:
            physical = min(len(cores), logical)
    except Exception:
        pass
    n = max(1, min(physical, CPU_THREADS_CAP))
    torch.set_num_threads(n)
    cfg = p_config()
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=P_XLNET))   # parameters only (CPU)
    p = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v)
         for k, v in model.state_dict().items()}
    del model
    B = int(os.environ.get("VILCO_CPU_BASELINE_CLIPS", "2"))
    steps = int(os.environ.get("VILCO_CPU_BASELINE_STEPS", "3"))
    vl = synth_batch(B, "cpu")
    mq_oracle.DROP = mq_oracle.DropRandom(dropout=0.1, droppath=0.1, xl=P_XLNET["dropout"], seed=0)
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass

    def step():
        for v in p.values():
            if v.is_floating_point():
                v.grad = None
        losses, _ = mq_oracle.forward_losses(p, cfg, vl)
        losses['final_loss'].backward()

    def report(dt, what):
        print(json.dumps({"value": B / dt, "unit": "clips/s", "cores": n, "kind": "port",
                          "sample": "%s of %d clips (T=2304, C=2304, config P, train-mode dropout 0.1 / droppath 0.1 / "
                                    "XLNet dropout 0.1), fp32, oracle/mq_oracle.py (CPU restatement of the reference), "
                                    "%.1f s per step, %d threads (host: %d physical cores / %d logical, %s)"
                                    % (what, B, dt, n, physical, logical, cpu)}), flush=True)
    t0 = time.time()
    step()                                            # warm-up (allocator, thread pool)
    report(time.time() - t0, "the warm-up fwd+bwd step only (timed steps did not fit the budget)")
    t0 = time.time()
    for i in range(steps):
        step()
        report((time.time() - t0) / (i + 1), "1 warm-up + %d timed fwd+bwd steps" % (i + 1))


def cpu_baseline(timeout_s=150):
    """bounded: runs in a child process (no GPU use); after `timeout_s` the child is stopped (its own PID) and the last
    line it printed is the result."""
    import subprocess
    import threading
    lines = []
    try:
        proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"],
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

        def reader():
            for l in proc.stdout:
                if l.startswith("{"):
                    lines.append(l)
        th = threading.Thread(target=reader, daemon=True)
        th.start()
        try:
            proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()
        th.join(timeout=5)
        return json.loads(lines[-1])
    except Exception as e:      # nothing finished (OOM on a small host, ...): report that instead of a number
        return {"value": None, "unit": "clips/s", "cores": min(os.cpu_count() or 1, CPU_THREADS_CAP), "kind": "port",
                "sample": "no oracle step finished within %d s (%s)" % (timeout_s, type(e).__name__)}


def gemm_profile(step_fn, n=3):
    """HIP-event timing of the MFMA GEMM kernels alone (gemm_gl_kernel: the fp16 x2 two-part products, round 5; gemm_pp_kernel:
    the single-part weight-gradient and bf16 products; all instantiations), on the stream they are
    launched on, over `n` extra steps: vilco_gemm brackets its main kernel with events (vilco_gemm_profile_*), the
    algorithmic FLOPs are counted at the ops.gemm call sites.  Also times the whole vilco_gemm calls (packs, kernel,
    split-K reduce) with torch events.  Returns a dict."""
    import ctypes
    from vilco_amd import ops, _lib
    lib = _lib.load()
    recs = []
    real_gemm, real_pack = ops.gemm, ops.pack

    def timed(fn, flop):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        recs.append((e0, e1, flop))
        return out

    mfma_flops = [0.0]          # MFMA work actually issued: x3 (fp16 x2 planes), x1 for the single-part weight-gradient products

    def gemm(A, B, Cc, M, N, K, *a, **k):
        batch = k.get("batch", (1, 1))
        fl = 2.0 * M * N * K * batch[0] * batch[1]
        prec = k.get("precision")
        prec = ops.get_precision() if prec is None else prec
        mfma_flops[0] += fl * {0: 3, 1: 1, 2: 6, 3: 3, 4: 1}[prec]
        return timed(lambda: real_gemm(A, B, Cc, M, N, K, *a, **k), fl)

    def pack(*a, **k):
        return timed(lambda: real_pack(*a, **k), 0.0)
    ops.gemm, ops.pack = gemm, pack
    try:
        _lib.check(lib.vilco_gemm_profile_begin())
        for _ in range(n):
            step_fn()
        torch.cuda.synchronize()
        ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
        _lib.check(lib.vilco_gemm_profile_end(ctypes.byref(ms), ctypes.byref(cnt)))
    finally:
        ops.gemm, ops.pack = real_gemm, real_pack
    flops = sum(f for _, _, f in recs)
    call_ms = sum(a.elapsed_time(b) for a, b, _ in recs)
    launches = int(cnt.value)
    return {"avg_launch_us": ms.value * 1e3 / launches, "tflops": flops / (ms.value * 1e-3) / 1e12,
            "launches_per_step": launches // n, "kernel_ms_per_step": ms.value / n,
            "gflop_per_launch": flops / launches / 1e9, "mfma_tflops": mfma_flops[0] / (ms.value * 1e-3) / 1e12,
            "mfma_per_product_avg": mfma_flops[0] / flops,
            "call_ms_per_step": call_ms / n, "call_tflops": flops / (call_ms * 1e-3) / 1e12}


def pmc_traffic():
    """HBM bytes per GEMM-kernel launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as the gfx950
    note in MI355X_MICROARCH.md prescribes, + WRITE_SIZE); None when the summary is absent."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):          # newest committed summary
        try:
            d = json.load(open(os.path.join(prof, name)))
            return (d.get("gemm_kernels") or d["gemm_pp_kernel"])["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def dryrun_parts(dev, rank):
    """VILCO_BENCH_DRYRUN=1 (tests/test_dist_cpu.py): the rank / collective protocol of this file with a small torch
    module on CPU + gloo in place of the HIP model, so that `bench.py --gpus N` is known to terminate under torchrun
    before it ever meets N GPUs."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 4))
    x = torch.randn(8, 16, generator=torch.Generator().manual_seed(rank))

    def fwd_bwd():
        model(x).pow(2).mean().backward()
    return model, fwd_bwd


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: N ranks as children of this process (one per GPU, rendezvous on
    127.0.0.1 at a free port), their stdout / stderr relayed, exit status = theirs.  Runs before any GPU call here."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env)
    try:
        rc = proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        rc = proc.wait()
    sys.exit(rc)


def main():
    if os.environ.get("VILCO_BENCH_WATCHDOG"):          # (diagnosis: dump every thread's stack after this many seconds, keep running)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["VILCO_BENCH_WATCHDOG"]), repeat=False, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="clips per GPU (mq_vilco.yaml: 2)")
    ap.add_argument("--precision", default="f16x2", choices=["split3", "split", "bf16", "f16x2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-targets", action="store_true", help="skip the target-kernel micro-benchmarks and the W / cfg1 side lines")
    ap.add_argument("--extra-batch", type=int, default=8,
                    help="also time a few steps at this many clips per GPU (reported beside the headline; 0 = skip)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N` (no launcher): start the N ranks as CHILD processes under torch.distributed.run --
        # the launch line of MQ/train_cl.sh:2 -- before this process has made any GPU call, relay their output (rank 0 prints
        # the JSON line) and leave with their status.  Never an exec: a parent that had touched the GPU must not be replaced.
        return self_launch(args.gpus)

    dry = os.environ.get("VILCO_BENCH_DRYRUN", "0") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # test hooks (tests / one-GPU boxes): VILCO_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and VILCO_BENCH_BACKEND=gloo carries the
    # device tensors over gloo (RCCL refuses two ranks on one device) -- the N > 1 code path of this file on a single GPU
    one_dev = os.environ.get("VILCO_BENCH_ONE_DEVICE", "0") == "1"
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(0 if one_dev else local_rank)
        dev = torch.device("cuda", 0 if one_dev else local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if dry else os.environ.get("VILCO_BENCH_BACKEND", "nccl"))   # "nccl" IS RCCL on ROCm
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    if dry:
        model, fwd_bwd = dryrun_parts(dev, rank)
    else:
        import vilco_amd
        import vilco_amd.modeling as vm
        from vilco_amd import ops
        vilco_amd._lib.load()                                  # no fallback: fail here if the .so is missing
        ops.set_precision(args.precision)
        cfg = p_config()
        torch.manual_seed(0)                                   # identical replicas ...
        model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=P_XLNET))
        model = model.to(dev).train()
        torch.manual_seed(1000 + rank)                         # ... different dropout / droppath draws and clips per rank
        batch = synth_batch(args.batch, dev, seed=rank)

        def fwd_bwd():
            model(batch, is_training=True)['final_loss'].backward()

    reducer = None
    if distributed:
        from vilco_amd.dist import GradReducer
        reducer = GradReducer(model)

    params = [p for p in model.parameters()] if not dry else None

    def eager_step():
        if params is not None:
            for p in params:                      # == model.zero_grad(set_to_none=True) without the module-tree walk (1.5 ms)
                p.grad = None
        else:
            model.zero_grad(set_to_none=True)
        if reducer is not None:
            reducer.begin()
        fwd_bwd()
        if reducer is not None:
            reducer.finish()

    # One rank per GPU, no gradient exchange: the step (forward + backward) is captured once as a hipGraph and replayed
    # (vilco_amd/graph.py) -- ~1500 launches enqueued by the runtime instead of by ~25 ms of Python.  Same kernels, same
    # arithmetic; dropout masks move with a device-side step word, stochastic-depth factors are re-drawn by every replay.
    # With the RCCL gradient exchange (N > 1) the step stays eager: the all-reduce is launched from autograd hooks while
    # backward is still running.  VILCO_BENCH_GRAPH=0 forces the eager step.  (Rounds 1-3; see below for N > 1 now.)
    graphed = None
    if not dry and os.environ.get("VILCO_BENCH_GRAPH", "1") != "0":
        from vilco_amd.graph import GraphedStep
        # N > 1 (round 4): the replayed step is the data-parallel step too.  The captured weight-gradient kernels write into
        # the reducer's bucket slots, the bucketed RCCL all-reduce runs right after the replay (GradReducer.reduce_now) and
        # averages in place.  VILCO_BENCH_GRAPH=0: eager launches, all-reduce from autograd hooks under backward.
        graphed = GraphedStep(model, None, eager_steps=2, reducer=reducer)

        def step():
            graphed(batch)
        if distributed:
            # Every rank must take the same path (replayed or eager) from the first timed step on: the capture is attempted here,
            # outside any step (a staged capture issues no collective), and the ranks agree on the outcome -- a capture that this
            # runtime refuses on ONE rank falls back to eager steps on ALL of them (all-reduce from autograd hooks) instead of
            # leaving the ranks waiting in different collectives.
            for _ in range(2):
                graphed(batch)                    # eager: builds the bucket plan (collectives inside, every rank alike)
            ok = torch.tensor([1 if graphed.try_capture(batch) else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                graphed.reset()
                graphed = None
                step = eager_step
    else:
        step = eager_step

    def fence():
        if distributed:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    # A cold process on a cold device runs the first ~40 steps 8 % slower (code objects loading, allocator growth, clock
    # ramp; same-box A/B, r02): settle for a fixed wall time before the W warm-up steps the contract asks for.  Untimed.
    if not dry:
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < float(os.environ.get("VILCO_BENCH_SETTLE_S", "3.0")):
            for _ in range(5):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms = dt / args.steps * 1e3
    clips_per_s = world * args.batch * args.steps / dt

    # Diagnostics of the gradient exchange (EVERY rank runs them: they contain collectives): each bucket's all-reduce
    # alone, and the same step without the exchange -- what the exchange costs, and how much of it backward hides.
    multi = None
    if reducer is not None:
        buckets = reducer.profile_buckets()
        reducer.enabled = False
        for _ in range(2):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(5):
            step()
        fence()
        local_ms = (time.perf_counter() - t1) / 5 * 1e3
        tot = sum(b["ms"] for b in buckets)
        eager_x_ms = None
        if graphed is not None:
            # the same data-parallel step launched eagerly (all-reduce from autograd hooks, overlapped with backward), for
            # comparison with the replayed one -- every rank runs it (collectives inside)
            reducer.enabled = True
            for _ in range(2):
                eager_step()
            fence()
            t1 = time.perf_counter()
            for _ in range(5):
                eager_step()
            fence()
            eager_x_ms = (time.perf_counter() - t1) / 5 * 1e3
            reducer.enabled = False
        staged = graphed is not None and any(e.get('seg_graphs') for e in graphed._graphs.values())
        multi = {"step_mode": ("backward replayed as %d hipGraphs (heads + losses, then one per backbone stage); every gradient bucket's "
                               "all-reduce launched behind the stage that completes it, under the stages that follow"
                               % max(1 + len(e.get('seg_graphs') or ()) for e in graphed._graphs.values())) if staged
                              else ("hipGraph replay + in-place bucketed all-reduce after the replay (no overlap with backward)"
                                    if graphed is not None else "eager launches, all-reduce from autograd hooks (overlapped with backward)"),
                 "eager_step_with_hook_exchange_ms": eager_x_ms,
                 "buckets": buckets, "exchange_ms_sum_of_buckets_alone": tot, "ms_per_step_without_exchange": local_ms,
                 "exposed_exchange_ms": ms - local_ms,
                 "overlap_frac": (1.0 - max(ms - local_ms, 0.0) / tot) if tot > 0 else None,
                 "gradient_bytes": sum(b["mb"] for b in buckets) * 2 ** 20,
                 "note": "bucketed RCCL all-reduce (AVG) launched from autograd hooks as buckets complete; the weight-gradient "
                         "kernels write into the bucket slots (no gather copy); busbw = algbw * 2 (n - 1) / n"}

    # Everything below is RANK-LOCAL (no collectives): rank 0 measures with the gradient exchange switched off while the
    # other ranks wait in the final barrier.  (Round 1 ran reducer steps on rank 0 only here, whose all-reduces had
    # no peer -- ADVICE r1.)
    if reducer is not None:
        reducer.enabled = False
    if rank == 0:
        out = {"metric": "clips/sec (fwd+bwd) MQ transformer T=2304 C=2304", "value": clips_per_s,
               "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": {"split3": "bf16 MFMA x3-part split, fp32 accumulate (fp32-equivalent)",
                         "split": "bf16 MFMA x2-part split, fp32 accumulate", "bf16": "bf16",
                         "f16x2": "fp16 MFMA on 2-part splits of power-of-two scaled fp32 operands (22 significant bits, "
                                  "3 MFMAs per product), fp32 accumulate: fp32-equivalent; weight-gradient products with "
                                  "contraction length >= 2048 use the leading fp16 parts only (11-bit operands, 1 MFMA; "
                                  "<= 3.5e-4 of the 3-MFMA gradients at this config, tests/test_fullsize_gpu.py)"}[args.precision],
               "data": "synthetic",
               "config": {"workload": "MQ ViLCo backbone config P: T=2304 Cin=2304 D=1024 H=16 arch(2,2,5) XLNet layer "
                                      "text L=77x768 22 classes, train mode with the reference's dropout 0.1 / droppath 0.1 "
                                      "/ XLNet dropout 0.1 (mq_vilco.yaml, xlnet_config_1024.json)",
                          "clips_per_gpu": args.batch, "global_batch": world * args.batch,
                          "parallelism": "dp%d" % world},
               "inputs": "resident in HBM before the timed region (the reference's step includes the H2D copy of the "
                         "42 MB batch, meta_archs.py:1178: ~1 ms on PCIe 5, not part of `value`)",
               "clips_per_s_per_gpu": clips_per_s / world,
               "model_mfma_frac": clips_per_s / world * GFLOP_PER_CLIP_FWD_BWD / 1e3 / PEAK_BF16_TFLOPS}
        if multi is not None:
            out["multi_gpu"] = multi
        if dry:
            step()                                               # a rank-local step: must not touch the process group
            out["dryrun"] = True
        else:
            out["step_mode"] = ("hipGraph replay of the captured forward + backward (vilco_amd/graph.py): %d capture(s), %d replays"
                                % (graphed.stats['captured'], graphed.stats['replayed'])) if graphed is not None else "eager launches"
            local_sections(out, args, model, step, eager_step, dev, ms, world, batch)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def side_config(name, dev, steps=10, warm=2):
    """fwd+bwd clips/s of another BASELINE / SURVEY 8d configuration, a few steps only (side line, not the headline):
    "W" = config P with D = 2304 (hd = 144, no XLNet layer; 9874 GFLOP/clip), "cfg1" = BASELINE configs[0] (T = 256,
    Cin = 512, D = 512, H = 4, XLNet layer, 53.4 GFLOP/clip).  Both run the fused attention kernels (head dims up to 160)."""
    import vilco_amd.modeling as vm
    from vilco_amd.core.config import make_config
    if name == "W":
        over = dict(dataset=dict(input_dim=2304, num_classes=22, max_seq_len=2304),
                    model=dict(embd_dim=2304, fpn_dim=2304, head_dim=2304, n_head=16, backbone_arch=(2, 2, 5), use_abs_pe=True,
                               use_cross_modal=True, n_txt_in=768, max_buffer_len_factor=1.0, use_xl=False),
                    train_cfg=dict(init_loss_norm=100, dropout=0.0, droppath=0.1))
        T, Cin, gflop, xl = 2304, 2304, 9874.0, None
    else:
        over = dict(dataset=dict(input_dim=512, num_classes=22, max_seq_len=256),
                    model=dict(embd_dim=512, fpn_dim=512, head_dim=512, n_head=4, backbone_arch=(2, 2, 5), use_abs_pe=True,
                               use_cross_modal=True, n_txt_in=768, max_buffer_len_factor=1.0, use_xl=True),
                    train_cfg=dict(init_loss_norm=100, dropout=0.0, droppath=0.1))
        T, Cin, gflop = 256, 512, 53.4
        xl = dict(P_XLNET, d_model=512, n_head=4, d_head=128, d_inner=1024, dropout=0.0)
    cfg = make_config(**over)['model']
    torch.manual_seed(0)
    kw = dict(cfg, xlnet_config=xl) if xl is not None else dict(cfg)
    model = vm.make_meta_arch('LocPointTransformer', **kw).to(dev).train()
    batch = synth_batch(2, dev, seed=0, T=T, Cin=Cin)
    if T < 2304:      # the synthetic segments of synth_batch end at 130.25 < 256: fine
        pass

    def eager_one():
        model.zero_grad(set_to_none=True)
        model(batch, is_training=True)['final_loss'].backward()
    use_graph = os.environ.get("VILCO_BENCH_GRAPH", "1") != "0"
    if use_graph:
        from vilco_amd.graph import GraphedStep
        gs = GraphedStep(model, None, eager_steps=1)

        def one():
            gs(batch)
    else:
        one = eager_one
    for _ in range(warm + 2):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    host = host_enqueue_ms(one, 3)
    eager_ms = None
    if use_graph:
        del gs
        for _ in range(2):
            eager_one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eager_one()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) / steps * 1e3
    del model
    torch.cuda.empty_cache()
    return {"config": name, "clips_per_gpu": 2, "ms_per_step": dt * 1e3, "clips_per_s": 2 / dt, "gflop_per_clip_fwd_bwd": gflop,
            "model_mfma_frac": 2 / dt * gflop / 1e3 / PEAK_BF16_TFLOPS, "steps": steps,
            "host_enqueue_ms": host, "eager_ms_per_step": eager_ms,
            "step_mode": "hipGraph replay" if use_graph else "eager launches",
            "note": "dropout 0 (droppath 0.1), %d timed steps after %d warm-up: a side line, not the headline workload" % (steps, warm)}


def host_enqueue_ms(fn, n=5):
    """host time to ISSUE one step (queue drained before each measurement, nothing waited for after it)"""
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return sorted(ts)[len(ts) // 2] * 1e3


def local_sections(out, args, model, step, eager_step, dev, ms, world, batch):
    """rank-0-only measurements after the timed region: host time, GEMM roofline, optimizer step, larger batch, CPU baseline"""
    out["host_enqueue_ms"] = host_enqueue_ms(step)
    if step is not eager_step:
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(8):
            eager_step()
        torch.cuda.synchronize()
        out["eager"] = {"ms_per_step": (time.perf_counter() - t1) / 8 * 1e3, "host_enqueue_ms": host_enqueue_ms(eager_step),
                        "note": "the same step launched kernel by kernel from Python (no graph)"}
    step = eager_step           # everything below instruments or re-times individual launches: eager
    gp = gemm_profile(step)
    mfma_per_product = {"f16x2": 3, "split3": 6, "split": 3, "bf16": 1}[args.precision]
    out["roofline"] = {"bound": "mfma", "kernel": "gemm_gl_kernel + gemm_pp_kernel (the MFMA GEMM family, all instantiations)", "achieved": gp["tflops"],
                       "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": gp["tflops"] / PEAK_BF16_TFLOPS,
                       "traffic": pmc_traffic(), "avg_launch_us": gp["avg_launch_us"],
                       "launches_per_step": gp["launches_per_step"],
                       "kernel_ms_per_step": gp["kernel_ms_per_step"],
                       "algorithmic_gflop_per_launch": gp["gflop_per_launch"],
                       "mfma_per_algorithmic_product": gp["mfma_per_product_avg"],
                       "mfma_issue_frac": gp["mfma_tflops"] / PEAK_BF16_TFLOPS,
                       "gemm_calls_ms_per_step_incl_pack_and_reduce": gp["call_ms_per_step"],
                       "gemm_calls_tflops_incl_pack_and_reduce": gp["call_tflops"]}
    # the optimizer step is reported separately (BASELINE.json metric = fwd+bwd): fused clip-norm + AdamW
    from vilco_amd.utils.train_utils import make_optimizer
    opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-4))
    step()
    opt.step(clip_grad_l2norm=1.0)                         # allocates state, builds the chunk plan
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        opt.step(clip_grad_l2norm=1.0)
    e1.record()
    torch.cuda.synchronize()
    n_par = sum(p.numel() for p in model.parameters() if p.grad is not None)
    opt_ms = e0.elapsed_time(e1) / 3
    # a whole training iteration, measured (not summed): zero_grad + fwd + bwd + clip + AdamW with lr > 0, so every weight
    # changes and its operand planes are re-packed in the next forward (in the headline loop weights stand still and their
    # planes are packed once).  The update kernel leaves max|w| per chunk, the re-pack takes its scale from there.
    use_graph = world == 1 and os.environ.get("VILCO_BENCH_GRAPH", "1") != "0"
    if use_graph:           # captured: graph 1 = forward + backward (weight re-packs inside), graph 2 = clip + AdamW
        from vilco_amd.graph import GraphedStep
        gtrain = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=1)

        def train_iter():
            gtrain(batch)
    else:
        def train_iter():
            step()
            opt.step(clip_grad_l2norm=1.0)
    for _ in range(4):
        train_iter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(8):
        train_iter()
    torch.cuda.synchronize()
    it_ms = (time.perf_counter() - t1) / 8 * 1e3
    it_host = host_enqueue_ms(train_iter)
    if use_graph:
        del gtrain
    out["optimizer_step"] = {"ms": opt_ms, "params_with_grad": n_par, "kind": "fused clip_grad_norm + AdamW",
                             "hbm_GBps": (32.0 * n_par) / (opt_ms * 1e-3) / 1e9,
                             "train_step_ms_incl_optimizer": ms + opt_ms,
                             "train_iteration_ms_measured": it_ms,
                             "train_iteration_clips_per_s": args.batch * 1e3 / it_ms,
                             "train_iteration_host_enqueue_ms": it_host,
                             "train_iteration": "zero_grad + fwd + bwd + clip_grad_norm + AdamW + weight re-pack, 8 iterations, wall "
                                                "clock" + (", replayed as two hipGraphs" if use_graph else ", eager launches")}
    if world == 1 and args.precision == "f16x2":
        # "strict": the same workload with EVERY product in the 22-bit two-part format (VILCO_DW_PRECISION=f16x2: the
        # weight-gradient products take 3 MFMAs too) and the per-iteration weight re-pack inside the loop -- what the
        # 1e-3-safe arithmetic costs without the single-part shortcut the headline takes
        from vilco_amd import ops
        keep = ops.dw_precision
        try:
            ops.dw_precision = None
            if use_graph:
                gfb = GraphedStep(model, None, eager_steps=1)
                gtr = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=1)
                fb, tr = (lambda: gfb(batch)), (lambda: gtr(batch))
            else:
                fb, tr = step, (lambda: (step(), opt.step(clip_grad_l2norm=1.0)))
            res = {}
            for name, fn in (("train_iteration_ms", tr), ("fwd_bwd_ms_static_weights", fb)):
                for _ in range(4):
                    fn()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(8):
                    fn()
                torch.cuda.synchronize()
                res[name] = (time.perf_counter() - t1) / 8 * 1e3
            res["fwd_bwd_ms_incl_weight_repack"] = res["train_iteration_ms"] - opt_ms
            res["clips_per_s_fwd_bwd"] = args.batch * 1e3 / res["fwd_bwd_ms_static_weights"]
            res["clips_per_s_fwd_bwd_incl_weight_repack"] = args.batch * 1e3 / res["fwd_bwd_ms_incl_weight_repack"]
            res["what"] = ("every matrix product on 2-part fp16 operands (3 MFMAs, 22 bits) incl. the weight gradients; "
                           "train_iteration = zero_grad + fwd + bwd + clip + AdamW with every weight re-packed each iteration")
            out["strict"] = res
            if use_graph:
                del gfb, gtr
        finally:
            ops.dw_precision = keep
    out["headline_note"] = ("the timed loop keeps the weights static (their operand planes are packed once); a training iteration "
                            "re-packs them: fwd + bwd incl. re-pack = train_iteration_ms_measured - optimizer ms = %.2f ms"
                            % (it_ms - opt_ms))
    if world == 1 and args.extra_batch and args.extra_batch != args.batch:
        # not the headline (the reference trains with 2 clips per GPU): shows what is launch-bound at batch 2
        del opt
        model.zero_grad(set_to_none=True)
        big = synth_batch(args.extra_batch, dev, seed=1)

        if use_graph:
            gbig = GraphedStep(model, None, eager_steps=1)

            def big_step():
                gbig(big)
        else:
            def big_step():
                model.zero_grad(set_to_none=True)
                model(big, is_training=True)['final_loss'].backward()
        for _ in range(3):
            big_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            big_step()
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - t1) / 5
        out["larger_batch"] = {"clips_per_gpu": args.extra_batch, "ms_per_step": dtb * 1e3,
                               "clips_per_s": args.extra_batch / dtb}
        if use_graph:
            del gbig
    if world == 1 and not args.no_targets and args.precision == "f16x2":
        # BASELINE.md 3.3 asks for a bf16 perf run beside the parity run: the SAME step with single-pass bf16 MFMA operands
        # (1 MFMA per product, 8-bit mantissas).  It does NOT meet the 1e-3 parity bar (DESIGN_LOG.md 3.1: median gradient
        # error 6e-2) and is not the headline.
        from vilco_amd import ops
        try:
            ops.set_precision("bf16")
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            dtb = (time.perf_counter() - t1) / 8
            out["bf16_perf_run"] = {"ms_per_step": dtb * 1e3, "clips_per_s": args.batch / dtb, "parity": "fails the 1e-3 bar "
                                    "(single bf16 pass, 2^-9 operands); reported because BASELINE.md 3.3 asks for it"}
        finally:
            ops.set_precision(args.precision)
    if world == 1 and not args.no_targets:
        # the two kernels the north star sets explicit targets on, and the other SURVEY 8d configurations
        del model
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_targets
        try:
            out["targets"] = {
                "qkv_pre_projection": {"target": ">= 60 % of 8 TB/s HBM on the fused LN -> depthwise conv x3 -> LN x3 at T = C = 2304",
                                       "measured": [bench_targets.qkv_pre_target(dev, 2), bench_targets.qkv_pre_target(dev, 8)]},
                "cross_attention": {"target": ">= 40 % MFMA utilisation on the cross-attention block (T' = 1152, L = 77, D = 1024)",
                                    "measured": [bench_targets.cross_attn_target(dev, 2), bench_targets.cross_attn_target(dev, 8)]}}
        except Exception as e:      # a side measurement must not take the headline line down
            out["targets"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["side_configs"] = []
        for name in ("cfg1", "W"):
            try:
                out["side_configs"].append(side_config(name, dev))
            except Exception as e:
                out["side_configs"].append({"config": name, "error": "%s: %s" % (type(e).__name__, e)})
    if world == 1 and not args.no_targets:
        # NMS timing leg: the reference's compiled CPU extension vs the HIP kernels on the same candidates
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            import nms_bench
            out["nms"] = nms_bench.nms_timing()
        except Exception as e:
            out["nms"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()


if __name__ == "__main__":
    main()
