"""Round 6: a heavier torch-only version of the reducer's replay (tools/lab/nullstream_torch_repro.py was exact): a graph of ~600 nodes
writes the step counter into 260 small "gradients" (pool memory, as p.grad of a captured step); per step 13 buckets are gathered with
_foreach_copy_, each followed by the event choreography of an asynchronous collective on a high-priority side stream, then a second
graph, then per bucket the wait + copy back.  No host wait.  argv: null|own [iters]"""
import sys
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "null"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
ctr = torch.zeros(1, device=dev)
big = torch.randn(3072, 3072, device=dev)
scratch = torch.empty(3072, 3072, device=dev)
NB, PER = 13, 20
sizes = [64 + 32 * (i % 7) for i in range(NB * PER)]
tmp = [torch.zeros(n, device=dev) for n in sizes]


def body(grads):
    ctr.add_(1)
    for i in range(NB * PER):
        if i % 20 == 0:
            torch.mm(big, big, out=scratch)
        tmp[i].copy_(ctr.expand(sizes[i]))
        grads.append(tmp[i] * 1.0)                 # a fresh tensor from the graph's pool, like a captured p.grad


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        body([])
torch.cuda.synchronize()
ctr.zero_()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
grads = []
with torch.cuda.graph(g1):
    body(grads)
with torch.cuda.graph(g2, pool=g1.pool()):
    for _ in range(40):
        torch.mm(big, big, out=scratch)
ctr.zero_()
torch.cuda.synchronize()
buckets = []
for b in range(NB):
    gs = grads[b * PER:(b + 1) * PER]
    flat = torch.zeros(sum(t.numel() for t in gs), device=dev)
    views, o = [], 0
    for t in gs:
        views.append(flat[o:o + t.numel()]); o += t.numel()
    buckets.append((gs, views))
side = torch.cuda.Stream(priority=-1)
own = torch.cuda.Stream() if mode == "own" else None
outs = []


def step():
    g1.replay()
    ends = []
    for gs, views in buckets:
        torch._foreach_copy_(views, gs)
        e = torch.cuda.Event(); e.record(torch.cuda.current_stream()); side.wait_event(e)
        end = torch.cuda.Event(); end.record(side); ends.append(end)
    g2.replay()
    for (gs, views), end in zip(buckets, ends):
        torch.cuda.current_stream().wait_event(end)
        torch._foreach_copy_(gs, views)
    outs.append(torch.stack([t[0] for t in grads]))


for it in range(iters):
    if own is not None:
        own.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(own):
            step()
        torch.cuda.current_stream().wait_stream(own)
    else:
        step()
torch.cuda.synchronize()
got = torch.stack(outs).cpu()
want = torch.arange(1, iters + 1, dtype=torch.float32)[:, None].expand_as(got)
badrows = (got != want).any(1).nonzero().flatten().tolist()
print("torch %s, stream %s, %d iterations, %d gradients: %d steps wrong%s" % (torch.__version__, mode, iters, got.shape[1], len(badrows),
      (" (first at step %d: %d tensors off, e.g. got %s want %s)" % (badrows[0], int((got[badrows[0]] != want[badrows[0]]).sum()),
       got[badrows[0]][got[badrows[0]] != want[badrows[0]]][:4].tolist(), float(want[badrows[0]][0]))) if badrows else ""))
