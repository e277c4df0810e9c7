"""world_size-2 gloo test of the gradient exchange (the N>1 path of bench.py / SURVEY.md 8e)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.unused = torch.nn.Linear(16, 16)      # never called: must be pruned from the plan
        self.b = torch.nn.Linear(16, 4)

        self.sometimes = torch.nn.Linear(16, 16)   # skipped by rank 1 in step 1: zero-filled, bucket still launched

    def forward(self, x, use_sometimes=True):
        h = torch.relu(self.a(x))
        if use_sometimes:
            h = h + self.sometimes(h)
        return self.b(h)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vilco_amd.dist import GradReducer
    torch.manual_seed(0)
    model = Tiny()
    red = GradReducer(model, bucket_mb=0.0001)      # tiny buckets -> several collectives
    outs = []
    for step in range(3):
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 * step + rank))
        model.zero_grad(set_to_none=True)
        red.begin()
        model(x, use_sometimes=not (rank == 1 and step == 1)).pow(2).sum().backward()
        local = {k: p.grad.clone().numpy() for k, p in model.named_parameters() if p.grad is not None}
        red.finish()
        outs.append((local, {k: p.grad.clone().numpy() for k, p in model.named_parameters() if p.grad is not None}))
    q.put((rank, outs, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(2):
        rank, outs, nb = q.get(timeout=120)
        res[rank] = outs
        assert nb >= 2
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(3):
        l0, r0 = res[0][step]
        l1, r1 = res[1][step]
        assert "unused.weight" not in r0
        for k in l0:
            want = (l0[k] + (l1[k] if k in l1 else 0.0)) / 2       # a rank without a gradient contributes zeros
            assert np.allclose(r0[k], want, atol=1e-6), k
            assert np.allclose(r1[k], want, atol=1e-6), k


def _replay_worker(rank, world, port, q):
    """the exchange protocol of a REPLAYED step (vilco_amd/graph.py): backward with the hooks silenced (`begin(hooks=False)`,
    what a capture does), a zero gradient for every planned parameter this rank's step did not reach, `reduce_now()` --
    against the hook-driven `finish()` of the eager step, same data"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vilco_amd import ops
    from vilco_amd.dist import GradReducer
    res = {}
    for mode in ("eager", "replay"):
        torch.manual_seed(0)
        model = Tiny()
        red = GradReducer(model, bucket_mb=0.0001)
        outs = []
        for step in range(3):
            x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 * step + rank))
            model.zero_grad(set_to_none=True)
            skip = rank == 1 and step >= 1
            if mode == "eager" or step == 0:              # (the plan is built by an eager step in both modes)
                red.begin()
                model(x, use_sometimes=not skip).pow(2).sum().backward()
                red.finish()
            else:
                red.begin(hooks=False)
                launched = len(red._pending)
                model(x, use_sometimes=not skip).pow(2).sum().backward()
                assert len(red._pending) == launched == 0          # no collective from the hooks
                red.end_capture()
                for p in red.planned():
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                held = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
                red.reduce_now()
                assert all(p.grad is held[k] for k, p in model.named_parameters() if p.grad is not None)   # averaged IN PLACE
            assert ops.grad_slot_provider is None                  # nothing may write into the buckets between steps
            outs.append({k: p.grad.clone().numpy() for k, p in model.named_parameters() if p.grad is not None})
        res[mode] = outs
        red.remove()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_replayed_step_exchange_equals_eager_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 29) % 2000)
    procs = [ctx.Process(target=_replay_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(3):
        e0, r0, e1, r1 = res[0]["eager"][step], res[0]["replay"][step], res[1]["eager"][step], res[1]["replay"][step]
        assert set(e0) == set(r0) == set(e1) == set(r1) and "sometimes.weight" in r1      # the same set is stepped on every rank
        for k in e0:
            assert np.array_equal(e0[k], r0[k]) and np.array_equal(e1[k], r1[k]), k           # replay protocol == eager, bit for bit
            assert np.array_equal(r0[k], r1[k]), k                                            # replicas hold the same average


def test_bench_protocol_two_ranks():
    """`bench.py --gpus 2` under torch.distributed.run terminates and rank 0 prints ONE JSON line: the rank /
    collective protocol of bench.py exercised on CPU + gloo (VILCO_BENCH_DRYRUN swaps the HIP model for a small torch
    module; everything else -- reducer, fences, the rank-local sections after the timed region -- is the real code)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VILCO_BENCH_DRYRUN="1")
    port = 29500 + ((os.getpid() + 7) % 2000)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=240,
                       env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["dryrun"] is True
    mg = out["multi_gpu"]                    # exchange diagnostics: per-bucket all-reduce time, step time without the exchange
    assert len(mg["buckets"]) >= 1 and all(b["ms"] > 0 and b["mb"] > 0 for b in mg["buckets"])
    assert mg["ms_per_step_without_exchange"] > 0 and mg["exchange_ms_sum_of_buckets_alone"] > 0


def test_bench_bare_gpus_form_self_launches():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE: bench.py starts its two ranks as child processes
    itself (before any GPU call), relays rank 0's single JSON line and leaves with the children's status."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VILCO_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0


class _FakeDetector(torch.nn.Module):
    """stands in for the model in the evaluator-format plumbing: deterministic segments per video id"""
    use_adapt = False

    def forward(self, video_list, task_id=0, is_training=False):
        out = []
        for v in video_list:
            key = sum((i + 1) * ord(c) for i, c in enumerate(v['video_id']))          # (str hashes differ between processes)
            n = 1 + key % 3
            g = torch.Generator().manual_seed(key)
            seg = torch.rand(n, 2, generator=g).sort(dim=1)[0]
            out.append({'video_id': v['video_id'], 'segments': seg, 'scores': torch.rand(n, generator=g),
                        'labels': torch.randint(0, 5, (n,), generator=g)})
        return out


def _val_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vilco_amd.utils.train_utils import collect_results_sharded
    batches = [[{'video_id': 'vid%02d' % i}] for i in range(7)]
    res = collect_results_sharded(batches, _FakeDetector(), rank=rank, world=world)
    q.put((rank, res['video-id'], res['t-start'].tolist(), res['label'].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_validation_equals_single_rank():
    """evaluator-format results gathered from two ranks = the single-rank pass (as a multiset of rows)"""
    from vilco_amd.utils.train_utils import collect_results, results_to_anet_json
    batches = [[{'video_id': 'vid%02d' % i}] for i in range(7)]
    single = collect_results(batches, _FakeDetector())
    assert sorted(single) == ['label', 'score', 't-end', 't-start', 'video-id'] and len(single['video-id']) == len(single['score'])
    js = results_to_anet_json(single)
    assert sorted(js) == ['external_data', 'results', 'version'] and set(js['results']) == set(single['video-id'])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 13) % 2000)
    procs = [ctx.Process(target=_val_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sorted(zip(single['video-id'], single['t-start'].tolist(), single['label'].tolist()))
    for rank, vids, starts, labels in got:
        assert sorted(zip(vids, starts, labels)) == want


class _LossModel(torch.nn.Module):
    """the `model(video_list) -> {'final_loss': ...}` contract of on_task_update"""
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(6, 3)
        self.reg_params = {}

    def forward(self, batch):
        return {'final_loss': self.lin(batch).pow(2).sum()}


def _ewc_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vilco_amd.cl_methods import regularizers
    torch.manual_seed(0)
    out = {}
    for kind in ('ewc', 'mas'):
        model = _LossModel()
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        loader = [torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * rank + i)) for i in range(2)]
        reg = regularizers.on_task_update(loader, 'cpu', opt, model, kind=kind, data_parallel=True)
        key = 'fisher' if kind == 'ewc' else 'importance'
        model.zero_grad(set_to_none=True)
        model(loader[-1])['final_loss'].backward()          # this rank's own last-batch gradient
        g = model.lin.weight.grad
        out[kind] = (reg[key][-1]['lin.weight'].numpy(), (g.pow(2) if kind == 'ewc' else g.abs()).numpy())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_ewc_mas_importance_is_averaged_over_ranks():
    """under data parallelism every rank takes the importance from the last batch of ITS shard; the penalty is
    applied after the gradient exchange, so the importances must agree -- on_task_update averages them over the group"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ewc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for kind in ('ewc', 'mas'):
        want = (res[0][kind][1] + res[1][kind][1]) / 2
        assert not np.allclose(res[0][kind][1], res[1][kind][1])          # the shards really differ
        assert np.allclose(res[0][kind][0], want, atol=1e-6) and np.allclose(res[1][kind][0], want, atol=1e-6)


def test_stage_bucket_schedule():
    """GraphedStep._bucket_schedule (the backward replayed in stages, vilco_amd/graph.py): a bucket may go out after stage k only
    if every gradient in it -- and in every bucket before it, the launch order being the plan order on all ranks -- was last
    touched in a stage <= k; the last stage releases everything."""
    import types
    from vilco_amd.graph import GraphedStep
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(6)]
    gs = GraphedStep.__new__(GraphedStep)
    gs.params = ps
    # plan: bucket 0 = {p0, p1}, bucket 1 = {p2}, bucket 2 = {p3, p4}; p5 is planned nowhere
    gs.reducer = types.SimpleNamespace(buckets=[{"params": [ps[0], ps[1]]}, {"params": [ps[2]]}, {"params": [ps[3], ps[4]]}])
    N = None
    marks = [
        [(10, 0), N, N, N, N, N],                       # stage 0 (graph 1) produced p0's gradient
        [(10, 0), (11, 0), (12, 0), N, N, N],           # stage 1: p1, p2
        [(10, 0), (11, 1), (12, 0), (13, 0), N, N],     # stage 2: p1 again (used twice), p3
        [(10, 0), (11, 1), (12, 0), (13, 0), N, (15, 0)],   # stage 3: only the unplanned p5; p4 never gets one (zero-filled)
    ]
    assert gs._bucket_schedule(marks) == [0, 0, 3, 3]     # bucket 0 waits for stage 2 (p1), bucket 1 behind it; p4 (no gradient) is ready from the start
    marks2 = [[(10, 0), (11, 0), N, N, N, N], [(10, 0), (11, 0), N, (13, 0), (14, 0), N], [(10, 0), (11, 0), (12, 0), (13, 0), (14, 0), N]]
    assert gs._bucket_schedule(marks2) == [1, 1, 3]       # bucket 2 is complete after stage 1 but must not overtake bucket 1
