"""Round 6: does aten::sum (multi-block reduction: staging buffer + semaphores cleared by cudaMemsetAsync inside the op) replay
correctly inside a captured graph on this torch / ROCm?  A device counter scales the input; every replay must return the new value.
Replayed on the default (null) stream and on a created stream."""
import torch
dev = torch.device("cuda:0")
print("torch", torch.__version__, "hip", torch.version.hip)
ctr = torch.zeros(1, device=dev)
own = torch.cuda.Stream()
for shape, dims in (((2, 2304, 1024), (0, 1)), ((2, 64, 1024), (0, 1)), ((4608, 1024), (0,)), ((2, 2304, 1024), (2,))):
    x = torch.randn(*shape, device=dev)
    ref = x.sum(dims)
    def body():
        ctr.add_(1)
        return (x * ctr).sum(dims)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    for where in ("null stream", "created stream"):
        ctr.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = body()
        res = []
        for k in range(4):
            if where == "null stream":
                g.replay()
            else:
                own.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(own):
                    g.replay()
                torch.cuda.current_stream().wait_stream(own)
            torch.cuda.synchronize()
            res.append(float((out.flatten() / ref.flatten()).median()))
        ok = all(abs(r - (k + 1)) < 1e-2 for k, r in enumerate(res))
        print("sum over dims %-6s of %-16s replayed on the %-14s: sum / reference per replay (want 1, 2, 3, 4) = %s  %s"
              % (dims, shape, where, [round(r, 3) for r in res], "ok" if ok else "<-- STALE SUM"))
