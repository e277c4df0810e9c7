# same-box A/B of two library builds on the step alone: tools/lab/libvilco_prev.so vs the current one, alternating, N rounds
for rep in $(seq 1 ${N:-4}); do
  for lib in prev cur; do
    if [ $lib = prev ]; then export VILCO_HIP_LIB=$PWD/tools/lab/libvilco_prev.so; else unset VILCO_HIP_LIB; fi
    python3 bench.py --no-cpu-baseline --no-targets --extra-batch 0 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$lib  step %.2f ms  gemm %.0f TF (%.2f ms/step)' % (d['ms_per_step'], r['achieved'], r['kernel_ms_per_step']))"
  done
done
unset VILCO_HIP_LIB
