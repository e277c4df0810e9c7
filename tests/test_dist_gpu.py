"""The data-parallel path on the device, as far as one GPU can show it: a ONE-rank RCCL group ("nccl" on ROCm) drives the
real GradReducer over the real HIP model -- bucket plan from the first backward, weight-gradient kernels writing into the
bucket slots (ops.grad_slot_provider), autograd hooks launching the all-reduces, p.grad re-pointed at the slots, the fused
optimizer's pointer tables following them, and (GraphedStep) the in-place exchange after a replayed backward.  With one
rank the average is the identity, so every variant must reproduce the plain single-process training run."""
import os

import pytest
import torch

from parity_util import build_hip_model, golden_inputs, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist
    if dist.is_initialized():
        yield None
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield None
    dist.destroy_process_group()


def _model(dev):
    gold = load_golden("noxl")
    model = build_hip_model(gold, dev).train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    model.loss_normalizer = 100.0
    return gold, model


def _train(dev, mode, steps=4):
    from vilco_amd import ops
    from vilco_amd.dist import GradReducer
    from vilco_amd.graph import GraphedStep
    from vilco_amd.utils.train_utils import make_optimizer
    gold, model = _model(dev)
    batch = golden_inputs(gold)
    opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-3))
    red = GradReducer(model, bucket_mb=0.05) if mode != "plain" else None
    graph = (GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=2, reducer=red, comm_in_graph=(mode == "graph_comm"),
                         segments=(mode == "graph"))
             if mode in ("graph", "graph_single", "graph_comm") else None)
    losses, in_slots = [], 0
    for it in range(steps):
        if graph is not None:
            l = graph(batch, task_id=gold['task_id'])
        else:
            for p in model.parameters():
                p.grad = None
            if red is not None:
                red.begin()
            l = model(batch, task_id=gold['task_id'], is_training=True)
            l['final_loss'].backward()
            if red is not None:
                if it > 0:       # from the second step on the plan exists: the dW kernels must have written into their slots
                    in_slots = sum(1 for b in red.buckets for p, v in zip(b["params"], b["views"])
                                   if p.grad is not None and p.grad.data_ptr() == v.data_ptr())
                red.finish()
                assert all(p.grad.data_ptr() == v.data_ptr() for b in red.buckets for p, v in zip(b["params"], b["views"]))
            opt.step(clip_grad_l2norm=1.0)
        losses.append(float(l['final_loss']))
    if red is not None:
        prof = red.profile_buckets(iters=1)
        assert len(prof) == len(red.buckets) >= 2 and all(b["ms"] > 0 for b in prof)
        red.remove()
        assert ops.grad_slot_provider is None
    if graph is not None:
        assert graph.stats['replayed'] == steps - 2, graph.stats
        nseg = [len(e.get('seg_graphs') or ()) for e in graph._graphs.values() if 'graph' in e]
        # "graph": the backward captured in stages (one graph per pyramid level of this model + the embedding / stem stage),
        # the buckets launched between their replays; the other modes: one graph
        assert (min(nseg) >= 2) if mode == "graph" else (max(nseg) == 0), (mode, nseg)
    return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, in_slots


def test_reducer_paths_reproduce_the_single_process_run(dev, one_rank_group):
    l0, s0, _ = _train(dev, "plain")
    l1, s1, in_slots = _train(dev, "hooks")
    assert in_slots >= 40, in_slots                 # the matrices' gradients were produced in place
    l2, s2, _ = _train(dev, "graph")                # backward replayed in stages, buckets launched between them (round 5)
    l3, s3, _ = _train(dev, "graph_single" if os.environ.get("VILCO_TEST_NO_COMM") else "graph_comm")   # the all-reduces captured inside graph 1 (or the fallback, if the runtime refuses)
    l4, s4, _ = _train(dev, "graph_single")         # one graph, the whole exchange after it
    for other_l, other_s in ((l1, s1), (l2, s2), (l3, s3), (l4, s4)):
        assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(other_l, l0)), (other_l, l0)
        for k in s0:
            # (key / key-norm biases: analytically zero gradients, 1e-12-level noise that Adam turns into +-lr steps)
            if s0[k].is_floating_point() and not k.endswith(('key_norm.bias', '.key.bias')):
                assert torch.equal(other_s[k], s0[k]) or rel_err(other_s[k], s0[k], 1e-7) < 1e-5, k


def test_p_config_staged_dp_replay_follows_eager_dp_training(dev, one_rank_group):
    """Round 6: the data-parallel step AT FULL SIZE -- backward replayed as stage graphs, every bucket's RCCL all-reduce launched behind
    the stage that completes it -- against the eager data-parallel step (all-reduce from autograd hooks), one rank, 24 training
    iterations over four rotating batches from the same initial state, dropout off: the same loss trajectory up to the staged path's
    only difference from eager arithmetic, the order in which the cut leaves accumulate their gradients (<= 2.4e-4 on the embedding
    convs), which the training dynamics amplify after a dozen iterations.  Rounds 4-5 compared the staged replay with the ONE-GRAPH replay at this size, never with eager steps."""
    import bench
    import vilco_amd.modeling as vm
    from vilco_amd.dist import GradReducer
    from vilco_amd.graph import GraphedStep
    from vilco_amd.utils.train_utils import make_optimizer
    cfg = bench.p_config(dropout=0.0, droppath=0.0)
    batches = [bench.synth_batch(2, dev, seed=s) for s in range(4)]
    runs = []
    for replay in (True, False):
        torch.manual_seed(0)
        model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet(dropout=0.0))).to(dev).train()
        opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-4))
        red = GradReducer(model)
        gs = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=2, reducer=red, enabled=replay, segments=True)
        losses = [gs(batches[it % 4])['final_loss'] for it in range(24)]
        torch.cuda.synchronize()
        if replay:
            assert gs.stats['replayed'] >= 20 and max(len(e.get('seg_graphs') or ()) for e in gs._graphs.values()) >= 2, gs.stats
        runs.append(torch.stack(losses).float().cpu())
        red.remove()
        del model, opt, gs, red
        torch.cuda.empty_cache()
    a, b = runs
    assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
    d = (a - b).abs() / b.abs()
    # measured (tools/lab/dp_soak.py and this test on three library builds): bit-equal over the first 6 iterations, <= 1e-6 up to the
    # 8th, then the rounding-order difference grows with the training dynamics by ~5x per iteration (2e-6 ... 2e-4 at iteration 11
    # depending on the build's GEMM plans, 4e-2 at 15); from iteration ~18 on every mode's loss spikes at this learning rate and
    # the comparison means nothing.  The one-graph replay and the plain replay are bit-equal to eager.
    assert float(d[:8].max()) < 1e-5 and float(d[:12].max()) < 2e-3 and float(d[:16].max()) < 0.2, d


# ---------------------------------------------------------------------------------------------------------------------
# Two REAL replicas on the one GPU of the box: two processes, both on cuda:0, gloo carrying the device tensors (RCCL refuses
# two ranks on one device).  Everything else is the production path: HIP model, FusedOptimizer, GradReducer, GraphedStep.
def _w2_worker(rank, world, port, mode, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vilco_amd.dist import GradReducer
        from vilco_amd.graph import GraphedStep
        from vilco_amd.utils.train_utils import make_optimizer
        from parity_util import cases
        dev = torch.device("cuda:0")
        gold, model = _model(dev)
        m = gold['overrides']
        batch = cases.video_list(m['dataset']['max_seq_len'], m['dataset']['input_dim'], m['model']['n_txt_in'], gold['L'],
                                 seed=40 + rank)                      # every replica trains on its own clips
        opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-3))
        red = GradReducer(model, bucket_mb=0.05)
        step = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=2, reducer=red, enabled=(mode == "graph"), segments=True)   # (opt-in: vilco_amd/graph.py _DP_SEGMENTS)
        losses = [float(step(batch, task_id=gold['task_id'])['final_loss']) for _ in range(5)]
        in_place = sum(1 for b in red.buckets for p, v in zip(b["params"], b["views"]) if p.grad is not None and p.grad.data_ptr() == v.data_ptr())
        stats = dict(step.stats, stage_graphs=max([len(e.get('seg_graphs') or ()) for e in step._graphs.values()] or [0]))
        q.put((rank, losses, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, stats, in_place))    # (numpy: by value)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _w2_run(mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + ((os.getpid() + (250 if mode == "graph" else 0)) % 500)
    procs = [ctx.Process(target=_w2_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(2):
        r = q.get(timeout=240)
        out[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


def test_world2_graph_replay_equals_eager_and_replicas_stay_identical(dev):
    eager, graph = _w2_run("eager"), _w2_run("graph")
    assert graph[0][2]['replayed'] == 3 and graph[1][2]['replayed'] == 3, (graph[0][2], graph[1][2])
    assert graph[0][2]['stage_graphs'] >= 2         # the replayed backward ran in stages with the exchange between them
    assert graph[0][3] >= 40                       # the captured weight-gradient kernels wrote into the bucket slots
    noise = ('key_norm.bias', '.key.bias')
    for rank in (0, 1):
        assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(graph[rank][0], eager[rank][0])), (graph[rank][0], eager[rank][0])
        for k, v in eager[rank][1].items():
            if v.dtype.kind == 'f' and not k.endswith(noise):
                g = graph[rank][1][k]
                assert (g == v).all() or rel_err(torch.from_numpy(g), torch.from_numpy(v), 1e-7) < 1e-5, (rank, k)
    assert eager[0][0] != eager[1][0]              # different clips per replica ...
    for k, v in graph[0][1].items():               # ... the same parameters after five averaged steps, bit for bit
        if v.dtype.kind == 'f':
            assert (v == graph[1][1][k]).all(), k
