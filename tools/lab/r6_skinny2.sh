#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r06_ab_skinny.txt; : > $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $out
for cfg in "VILCO_GEMM_SKINNY=0" "VILCO_GEMM_SKINNY=1" "VILCO_GEMM_SKINNY=0" "VILCO_GEMM_SKINNY=1"; do
  env $cfg timeout 600 python tools/lab/rtflags_cfg1.py "$cfg" >> $out 2>/dev/null
done
bash tools/ab_bench.sh "VILCO_GEMM_SKINNY=0" "VILCO_GEMM_SKINNY=1" "VILCO_GEMM_SKINNY=0" "VILCO_GEMM_SKINNY=1" >> $out 2>&1
cat $out
