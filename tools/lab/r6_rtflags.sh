#!/bin/bash
# HIP runtime graph knobs (libamdhip64: DEBUG_CLR_GRAPH_PACKET_CAPTURE, DEBUG_HIP_FORCE_GRAPH_QUEUES), one box: cfg1 (~1100 tiny
# nodes: the per-node floor) and the P step under each setting.
mkdir -p gpurun_out
out=gpurun_out/r06_rtflags.txt; : > $out
for cfg in "A=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" \
           "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "A=1"; do
  env $cfg timeout 600 python tools/lab/rtflags_cfg1.py "$cfg" >> $out 2>>gpurun_out/r06_rtflags.err
  o=$(env $cfg timeout 900 python bench.py --no-cpu-baseline --no-targets --extra-batch 0 2>>gpurun_out/r06_rtflags.err)
  echo "$o" | python -c "
import json,sys
d=json.load(sys.stdin)
print('%-42s P step %.2f ms  %.1f clips/s  train-iter %.2f ms' % (sys.argv[1], d['ms_per_step'], d['value'], d['optimizer_step']['train_iteration_ms_measured']))" "$cfg" >> $out
done
cat $out
