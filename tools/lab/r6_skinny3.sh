#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r06_skinny_bm16.txt; : > $out
for b in 32 160 320 640; do
  echo "== 16-row workgroups up to M = $b" >> $out
  VILCO_GEMM_SKINNY_BM16=$b timeout 600 python tools/lab/skinny_ab.py 2>&1 | grep -v amdgpu | grep "x   512 x   512\|x  1024 x   768" >> $out
  for r in 1 2; do VILCO_GEMM_SKINNY_BM16=$b timeout 600 python tools/lab/rtflags_cfg1.py "BM16=$b" >> $out 2>/dev/null; done
done
cat $out
