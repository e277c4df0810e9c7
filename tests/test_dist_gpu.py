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
    graph = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=2, reducer=red) if mode == "graph" else None
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
    return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, in_slots


def test_reducer_paths_reproduce_the_single_process_run(dev, one_rank_group):
    l0, s0, _ = _train(dev, "plain")
    l1, s1, in_slots = _train(dev, "hooks")
    assert in_slots >= 40, in_slots                 # the matrices' gradients were produced in place
    l2, s2, _ = _train(dev, "graph")
    for other_l, other_s in ((l1, s1), (l2, s2)):
        assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(other_l, l0)), (other_l, l0)
        for k in s0:
            # (key / key-norm biases: analytically zero gradients, 1e-12-level noise that Adam turns into +-lr steps)
            if s0[k].is_floating_point() and not k.endswith(('key_norm.bias', '.key.bias')):
                assert torch.equal(other_s[k], s0[k]) or rel_err(other_s[k], s0[k], 1e-7) < 1e-5, k
