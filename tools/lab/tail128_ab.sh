# the two-height launch (one round of 192-row tiles + a round of 128-row tiles) vs two rounds of 192-row tiles, operands packed
for shp in "9082 1024 3072 NT" "4608 2048 1024 NT" "2304 4096 1024 NT" "9216 1024 1024 NT"; do
  for t in 0 1; do
    echo -n "TAIL128=$t  "; VILCO_GEMM_TAIL128=$t python3 tools/gemm_one.py f16x2 $shp 2>&1 | grep -v amdgpu
  done
done
for t in 0 1; do echo TAIL128=$t; VILCO_GEMM_TAIL128=$t python3 bench.py --no-cpu-baseline --no-targets --extra-batch 0 2>/dev/null | tail -1 | cut -c1-200; done
