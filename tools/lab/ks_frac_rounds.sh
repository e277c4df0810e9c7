# shapes whose tile count is between one and two rounds of 256 CUs: unsplit vs two K splits (kernel + split finish, operands packed)
for shp in "9082 1024 3072 NT" "4608 2048 1024 NT" "2304 4096 1024 NT" "4608 1024 1024 NT" "4608 4096 1024 NT"; do
  for ks in 0 2; do
    echo -n "KS=$ks  "; VILCO_GEMM_KS=$ks python3 tools/gemm_one.py f16x2 $shp 2>&1 | grep -v amdgpu
  done
done
