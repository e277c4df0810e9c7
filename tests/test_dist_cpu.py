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

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


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
        model(x).pow(2).sum().backward()
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
            want = (l0[k] + l1[k]) / 2
            assert np.allclose(r0[k], want, atol=1e-6), k
            assert np.allclose(r1[k], want, atol=1e-6), k
