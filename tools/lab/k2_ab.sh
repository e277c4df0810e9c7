#!/bin/bash
# single-part (precision 4) products: two K-steps per barrier interval (default) vs one (VILCO_GEMM_K2=0), same box
for rep in 1 2; do
for k in 1 0; do
  export VILCO_GEMM_K2=$k
  echo "== VILCO_GEMM_K2=$k"
  for sh in "1024 1024 4608 TN" "1024 4096 4608 TN" "4096 1024 4608 TN" "1024 3072 4608 TN" "1024 1024 2304 TN" "1024 2048 4608 TN"; do
    python3 tools/gemm_one.py f16x2 $sh 4 2>/dev/null | tail -1
  done
done
done
