# A/B of two library builds on the same box: tools/lab/libvilco_prev.so vs the current one
for rep in 1 2; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export VILCO_HIP_LIB=$PWD/tools/lab/libvilco_prev.so; else unset VILCO_HIP_LIB; fi
    echo "== $lib"; python3 tools/gemm_one_time.py
  done
done
unset VILCO_HIP_LIB
