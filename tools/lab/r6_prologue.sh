#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r06_prologue_stamps.txt; : > $out
export VILCO_GEMM_SKINNY=0
for shp in "4608 1024 1024 NT" "4608 1024 1024 NN" "288 1024 1024 NT" "1024 1024 4608 TN"; do
  echo "== $shp" >> $out
  python tools/lab/gl_stamps.py $shp 2>&1 | grep "group\|prologue" >> $out
done
cat $out
