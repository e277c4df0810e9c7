#!/bin/bash
# staging loads two K-steps ahead (tools/lab/libvilco_d2.so: -DVILCO_GEMM_DEPTH2=1) vs one: isolated shapes, then the replayed P step
VILCO_HIP_LIB=$PWD/tools/lab/libvilco_d2.so timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or linear or conv3" 2>&1 | tail -2
for rep in 1 2; do
for v in cur d2; do
  if [ $v = cur ]; then unset VILCO_HIP_LIB; else export VILCO_HIP_LIB=$PWD/tools/lab/libvilco_d2.so; fi
  echo "== $v"
  for sh in "4608 1024 1024 NT" "4608 1024 1024 NN" "4608 4096 1024 NT" "4608 1024 4096 NT" "9082 1024 3072 NT"; do
    python3 tools/gemm_one.py f16x2 $sh 2>/dev/null | tail -1
  done
  PROBE_SKIP_EAGER=1 PROBE_FB_ONLY=1 timeout 300 python tools/graph_probe.py P 25 2>&1 | grep -E "graph fwd|rror" | head -2
done
done
