#!/bin/bash
# same-box comparison of the GEMM variants built by gemm_variants.sh on the step's main shapes (kernel only, operands packed)
for rep in 1 2; do
for v in cur "$@"; do
  if [ $v = cur ]; then unset VILCO_HIP_LIB; else export VILCO_HIP_LIB=$PWD/tools/lab/libvilco_v$v.so; fi
  echo "== variant $v"
  for sh in "4608 1024 1024 NT" "4608 1024 1024 NN" "4608 4096 1024 NT" "4608 1024 4096 NT" "9216 1024 1024 NT" "1024 4096 4608 TN"; do
    python3 tools/gemm_one.py f16x2 $sh 2>/dev/null | tail -1
  done
done
done
