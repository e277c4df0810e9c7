"""Where does a replayed data-parallel step spend its time when two replicas share ONE GPU over gloo (the only multi-rank setting a
1-GPU box allows)?  Splits the step into: graph replay (device-synchronised), reduce_now per bucket (launch / wait / copy back)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import vilco_amd.modeling as vm
    from vilco_amd.dist import GradReducer
    from vilco_amd.graph import GraphedStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    batch = bench.synth_batch(2, dev, seed=rank)
    red = GradReducer(model)
    g = GraphedStep(model, None, eager_steps=2, reducer=red)
    sync = lambda: torch.cuda.synchronize()
    for _ in range(4):
        g(batch); sync()
    dist.barrier()
    # instrument
    real_replay = GraphedStep._replay
    t = {"replay_graph": 0.0, "reduce_now": 0.0, "launch": 0.0, "wait": 0.0}
    real_reduce, real_launch = red.reduce_now, red._launch
    nosync = os.environ.get("PROBE_SYNC", "1") == "0"
    def timed_reduce():
        if nosync:
            t0 = time.perf_counter(); real_reduce(); t["reduce_now"] += time.perf_counter() - t0
            return
        sync(); t0 = time.perf_counter(); real_reduce(); sync(); t["reduce_now"] += time.perf_counter() - t0
    def timed_launch(b):
        t0 = time.perf_counter(); real_launch(b); t["launch"] += time.perf_counter() - t0
    red.reduce_now, red._launch = timed_reduce, timed_launch
    n = 3
    sync(); dist.barrier(); t0 = time.perf_counter()
    for _ in range(n):
        g(batch)
    sync(); total = time.perf_counter() - t0
    if rank == 0:
        print("PROBE_SYNC=%s OMP_NUM_THREADS=%s" % (os.environ.get("PROBE_SYNC", "1"), os.environ.get("OMP_NUM_THREADS")))
        print("2 ranks on one GPU, gloo: step %.1f ms | reduce_now %.1f ms (of which launching the %d all-reduces %.1f ms) | rest (replay + copies) %.1f ms | stats %s"
              % (total / n * 1e3, t["reduce_now"] / n * 1e3, len(red.buckets), t["launch"] / n * 1e3, (total - t["reduce_now"]) / n * 1e3, g.stats), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)
