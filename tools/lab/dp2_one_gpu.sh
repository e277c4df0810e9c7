# two replicas of the P step on ONE GPU over gloo (the only multi-rank setting a 1-GPU box allows): the N > 1 path of bench.py with
# the backward replayed in stages (default) and as one graph (VILCO_DP_SEGMENTS=0) -> gpurun_out/dp2_*.json
export VILCO_BENCH_ONE_DEVICE=1 VILCO_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for seg in 1 0; do
  VILCO_DP_SEGMENTS=$seg timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29511 + seg)) \
    bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-targets --extra-batch 0 > gpurun_out/dp2_seg$seg.json 2> gpurun_out/dp2_seg$seg.err
  echo "segments=$seg rc=$?"; tail -1 gpurun_out/dp2_seg$seg.json | cut -c1-300; tail -3 gpurun_out/dp2_seg$seg.err
done
