"""Round 6: the stale tensors of the null-stream hazard are the two gradients produced by `aten::sum` inside the captured step (XLNet's
r_w_bias / r_r_bias: the broadcast add's backward).  Torch only: a graph computes s = (x * ctr).sum((0, 1)) over [2, 2304, 1024] -- ATen's
multi-block reduction: staging buffer + semaphores allocated and cleared INSIDE the capture -- into a pool tensor; per step: replay,
gather copy, the asynchronous collective's event choreography on a high-priority side stream, a second graph, wait, copy back.
argv: null|own [iters] [nsums]"""
import sys
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "null"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nsums = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
ctr = torch.zeros(1, device=dev)
x = torch.ones(2, 2304, 1024, device=dev) / 4608.0
big = torch.randn(3072, 3072, device=dev)
scratch = torch.empty(3072, 3072, device=dev)


def body(out):
    ctr.add_(1)
    for i in range(nsums):
        for _ in range(6):
            torch.mm(big, big, out=scratch)
        out.append((x * ctr).sum((0, 1)))             # [1024], every element == ctr


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
    for _ in range(30):
        torch.mm(big, big, out=scratch)
ctr.zero_()
torch.cuda.synchronize()
flat = torch.zeros(1024 * nsums, device=dev)
views = [flat[i * 1024:(i + 1) * 1024] for i in range(nsums)]
side = torch.cuda.Stream(priority=-1)
own = torch.cuda.Stream() if mode == "own" else None
outs = []


def step():
    g1.replay()
    torch._foreach_copy_(views, grads)
    e = torch.cuda.Event(); e.record(torch.cuda.current_stream()); side.wait_event(e)
    end = torch.cuda.Event(); end.record(side)
    g2.replay()
    torch.cuda.current_stream().wait_event(end)
    torch._foreach_copy_(grads, views)
    outs.append(torch.stack([t[5] for t in grads]))


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
bad = ((got - want).abs() > 1e-3 * want).any(1).nonzero().flatten().tolist()
print("torch %s, stream %s, %d iterations, %d sums: %d steps wrong%s" % (torch.__version__, mode, iters, nsums, len(bad),
      (" (first at step %d: got %s want %.0f)" % (bad[0], got[bad[0]].tolist(), float(want[bad[0]][0]))) if bad else ""))
