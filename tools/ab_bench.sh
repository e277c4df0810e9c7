#!/bin/bash
# A/B of environment toggles on ONE box: tools/ab_bench.sh "VAR=a" "VAR=b" ...   (each argument = env assignments for one run)
for cfg in "$@"; do
  out=$(env $cfg python bench.py --no-cpu-baseline --no-targets --extra-batch 0 2>/dev/null)
  echo "$out" | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('%-40s step %.2f ms  %.1f clips/s  gemm %.0f TF (%.2f ms/step, calls %.2f ms)' % (sys.argv[1], d['ms_per_step'], d['value'], r['achieved'], r['kernel_ms_per_step'], r['gemm_calls_ms_per_step_incl_pack_and_reduce']))" "$cfg"
done
