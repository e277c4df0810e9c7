#!/bin/bash
# previous build (tools/lab/libvilco_vhead.so) vs the current one on the step's GEMM shapes, same box
for rep in 1 2; do
for v in cur head; do
  if [ $v = cur ]; then unset VILCO_HIP_LIB; else export VILCO_HIP_LIB=$PWD/tools/lab/libvilco_vhead.so; fi
  echo "== $v"
  for sh in "4608 1024 1024 NT" "4608 1024 1024 NN" "4608 4096 1024 NT" "4608 1024 4096 NT" "9216 1024 1024 NT" "2304 1024 1024 NT"; do
    python3 tools/gemm_one.py f16x2 $sh 2>/dev/null | tail -1
  done
  for sh in "1024 1024 4608 TN" "1024 4096 4608 TN"; do
    python3 tools/gemm_one.py f16x2 $sh 4 2>/dev/null | tail -1
  done
done
done
