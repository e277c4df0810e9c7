"""Is work queued right behind a hipGraph launch ordered after the WHOLE graph?  (in the staged data-parallel replay it
was not.)  A graph with a forked branch writes `flag = step` at its very end; right behind the replay (a) an eager kernel on the same
stream and (b) another stream through an event read the flag."""
import torch
dev = torch.device("cuda:0")
x = torch.randn(4096, 4096, device=dev); y = torch.randn(4096, 4096, device=dev)
step = torch.zeros(1, device=dev); flag = torch.zeros(1, device=dev); acc = torch.zeros(4096, 4096, device=dev)
side, other = torch.cuda.Stream(), torch.cuda.Stream()
def body():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        z = x
        for _ in range(6): z = (z @ y) * 1e-3
    w = x
    for _ in range(3): w = (w @ y) * 1e-3
    cur.wait_stream(side)
    acc.copy_(w + z)
    flag.copy_(step)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    bad_a = bad_b = 0
    n = 300
    snaps_a, snaps_b = [], []
    for i in range(1, n + 1):
        step.fill_(float(i))
        g.replay()
        snaps_a.append(flag.clone())                       # (a) eager kernel right behind the launch, same stream
        ev = torch.cuda.Event(); ev.record(s)
        other.wait_event(ev)
        with torch.cuda.stream(other):
            snaps_b.append(flag.clone())                   # (b) another stream behind an event recorded after the launch
    torch.cuda.synchronize()
    bad_a = sum(int(float(t) != float(i)) for i, t in enumerate(snaps_a, 1))
    bad_b = sum(int(float(t) != float(i)) for i, t in enumerate(snaps_b, 1))
print("replays %d: stale reads by the eager kernel behind the launch %d, by the other stream behind an event %d" % (n, bad_a, bad_b))

# (c) the other direction: does a graph launch wait for EAGER work queued before it on the same stream?  A long eager chain ends by
# writing `val = i`; the graph's first node copies `val` (main branch) and a forked branch copies it too.
val = torch.zeros(1, device=dev); snap_main = torch.zeros(1, device=dev); snap_side = torch.zeros(1, device=dev)
def body2():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        snap_side.copy_(val)
        z = x
        for _ in range(2): z = (z @ y) * 1e-3
    snap_main.copy_(val)
    w = (x @ y) * 1e-3
    cur.wait_stream(side)
    acc.copy_(w + z)
with torch.cuda.stream(s):
    for _ in range(2): body2()
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        body2()
    res_m, res_s = [], []
    for i in range(1, 201):
        t = x
        for _ in range(4): t = (t @ y) * 1e-3            # ~1 ms of eager work ...
        val.fill_(float(i)); val.add_(t[0, 0] * 0)          # ... whose last kernels set val = i
        g2.replay()
        res_m.append(snap_main.clone()); res_s.append(snap_side.clone())
    torch.cuda.synchronize()
    bm = sum(int(float(a) != float(i)) for i, a in enumerate(res_m, 1)); bs = sum(int(float(a) != float(i)) for i, a in enumerate(res_s, 1))
print("eager work before the launch, 200 replays: stale reads by the graph's first node %d, by its forked branch's first node %d" % (bm, bs))

# (d) the same two directions under DEEP host run-ahead: nine launches of a 200-node graph per iteration with eager kernels between
# them, no synchronisation for many iterations (the staged data-parallel replay's pattern)
import os
NIT, NG = int(os.environ.get("NIT", "80")), 9
val2 = torch.zeros(1, device=dev); first = torch.zeros(1, device=dev); last = torch.zeros(1, device=dev)
buf = torch.zeros(1 << 16, device=dev)
def body3():
    first.copy_(val2)
    t = buf
    for _ in range(200): t = t * 1.0001 + 1.0
    buf.copy_(t * 0)
    last.copy_(val2 + buf[0])
with torch.cuda.stream(s):
    for _ in range(2): body3()
    torch.cuda.synchronize()
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, stream=s):
        body3()
    A, Bv, want = [], [], []
    for i in range(NIT):
        for k in range(NG):
            v = float(i * 16 + k + 1)
            val2.fill_(v)                                  # eager, before the launch
            g3.replay()
            A.append(first.clone()); Bv.append(last.clone()); want.append(v)      # eager, behind the launch
    torch.cuda.synchronize()
    ba = sum(int(float(a) != w) for a, w in zip(A, want)); bb = sum(int(float(b) != w) for b, w in zip(Bv, want))
    firstbad = next((i for i, (a, b, w) in enumerate(zip(A, Bv, want)) if float(a) != w or float(b) != w), None)
print("deep run-ahead, %d launches: graph's first node read a stale eager value %d times, eager kernel behind the launch read a stale graph value %d times (first at launch %s)" % (len(want), ba, bb, firstbad))

# (e) as (c), with a small H2D copy from pinned memory (an SDMA transfer) between the eager work and the launch -- the pattern of a
# training step: ... eager kernels of the previous step, prepare()'s pinned H2D copies, graph launch
pin = torch.zeros(8, dtype=torch.int32).pin_memory(); dst_small = torch.zeros(8, dtype=torch.int32, device=dev)
with torch.cuda.stream(s):
    res_m, res_s = [], []
    for i in range(1, 201):
        t = x
        for _ in range(4): t = (t @ y) * 1e-3
        val.fill_(float(i)); val.add_(t[0, 0] * 0)
        pin.fill_(i)
        dst_small.copy_(pin, non_blocking=True)             # SDMA H2D, queued behind the eager kernels
        g2.replay()
        res_m.append(snap_main.clone()); res_s.append(snap_side.clone())
    torch.cuda.synchronize()
    bm = sum(int(float(a) != float(i)) for i, a in enumerate(res_m, 1)); bs = sum(int(float(a) != float(i)) for i, a in enumerate(res_s, 1))
print("eager work, then a pinned H2D copy, then the launch, 200 replays: stale reads by the graph's first node %d, by its forked branch %d" % (bm, bs))
