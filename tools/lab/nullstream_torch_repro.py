"""Round 6: the null-stream hazard reduced to torch alone?  A captured graph (optionally with a forked branch, as the captured training
step has) writes a step counter into a small tensor; between replays the reducer's choreography runs with torch objects only -- gather
copy, an event of the current stream, a high-priority side stream that waits for it and records an end event (nothing runs there),
a second graph, the wait for the end event, the copy back -- with the host running ahead.  argv: null|own  [iters] [fork 0|1]"""
import sys
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "null"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
fork = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
dev = torch.device("cuda:0")
ctr = torch.zeros(1, device=dev)
small = torch.zeros(1024, device=dev)
small_b = torch.zeros(1024, device=dev)
big = torch.randn(2048, 2048, device=dev)
scratch = torch.empty(2048, 2048, device=dev)
scratch2 = torch.empty(2048, 2048, device=dev)
side_cap = torch.cuda.Stream()


def body():
    ctr.add_(1)
    if fork:
        main = torch.cuda.current_stream()
        side_cap.wait_stream(main)
        with torch.cuda.stream(side_cap):
            torch.mm(big, big, out=scratch2)
            small_b.copy_(ctr.expand(1024))
    for _ in range(6):
        torch.mm(big, big, out=scratch)
    small.copy_(ctr.expand(1024))
    if fork:
        torch.cuda.current_stream().wait_stream(side_cap)


def body2():
    for _ in range(3):
        torch.mm(big, big, out=scratch)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        body(); body2()
torch.cuda.synchronize()
ctr.zero_()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    body()
with torch.cuda.graph(g2, pool=g1.pool()):
    body2()
ctr.zero_()
torch.cuda.synchronize()
bucket = torch.zeros(2048, device=dev)
side = torch.cuda.Stream(priority=-1)
own = torch.cuda.Stream() if mode == "own" else None
outs = []


def step():
    g1.replay()
    torch._foreach_copy_([bucket[:1024], bucket[1024:]], [small, small_b])          # gather
    e = torch.cuda.Event()
    e.record(torch.cuda.current_stream())
    side.wait_event(e)
    end = torch.cuda.Event()
    end.record(side)
    g2.replay()
    torch.cuda.current_stream().wait_event(end)
    torch._foreach_copy_([small, small_b], [bucket[:1024], bucket[1024:]])          # copy back
    outs.append(torch.stack([small[0], small_b[0]]))


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
want = torch.arange(1, iters + 1, dtype=torch.float32)[:, None].expand(iters, 2)
bad = (got != want).any(1).nonzero().flatten().tolist()
print("torch %s, stream %s, fork %s, %d iterations: %d wrong%s" % (torch.__version__, mode, fork, iters, len(bad),
      (" (first at %d: got %s want %s)" % (bad[0], got[bad[0]].tolist(), want[bad[0]].tolist())) if bad else ""))
