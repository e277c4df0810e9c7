#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r06_epi_stamps2.txt; : > $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $out
export VILCO_GEMM_SKINNY=0
for shp in "4608 1024 1024 NT" "4608 4096 1024 NT"; do
  for epi in "" "bias" "bias,res" "bias,gelu" "bias,res,amax"; do
    echo "== $shp EPI=$epi" >> $out
    EPI=$epi python tools/lab/gl_stamps.py $shp 2>&1 | grep "group" >> $out
  done
done
unset VILCO_GEMM_SKINNY
bash tools/lab/ab_lib.sh >> $out 2>&1
cat $out
