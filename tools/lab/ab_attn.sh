# attention kernels (incl. their packs) at T = 2304 / 1152: previous library build vs current, same box
for rep in 1 2; do
  echo prev; VILCO_HIP_LIB=$PWD/tools/lab/libvilco_prev.so python tools/attn_bench.py f16x2 2>/dev/null
  echo cur; python tools/attn_bench.py f16x2 2>/dev/null
done
