#!/bin/bash
# epilogue cycles of gemm_gl_kernel (lab build, in-kernel stamps) with and without fused bias / residual / GELU + pre-activation
mkdir -p gpurun_out; out=gpurun_out/r06_epi_stamps.txt; : > $out
export VILCO_GEMM_SKINNY=0
for shp in "4608 1024 1024 NT" "4608 4096 1024 NT" "288 1024 1024 NT"; do
  for epi in "" "bias" "bias,res" "bias,gelu" "bias,res,amax"; do
    echo "== $shp EPI=$epi" >> $out
    EPI=$epi python tools/lab/gl_stamps.py $shp 2>&1 | grep "group" >> $out
  done
done
cat $out
