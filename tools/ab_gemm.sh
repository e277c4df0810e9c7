# A/B of two library builds on the same box: tools/lab/libvilco_prev.so (build it first: make -C vilco_amd/csrc EXTRA=-D...
# and copy) vs the current one.  Per-shape GEMM kernel time of a P step (tools/gemm_shapes.py) and the bench step time.
for rep in 1 2; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export VILCO_HIP_LIB=$PWD/tools/lab/libvilco_prev.so; else unset VILCO_HIP_LIB; fi
    echo "== $lib"; python3 tools/gemm_shapes.py 3 2>/dev/null | head -${ROWS:-12}
    python3 bench.py --no-cpu-baseline --no-targets --extra-batch 0 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('step %.2f ms  gemm %.0f TF (%.2f ms/step)' % (d['ms_per_step'], r['achieved'], r['kernel_ms_per_step']))"
  done
done
unset VILCO_HIP_LIB
