# split count on the weight-gradient shapes (TN) and small-M forward shapes
for shape in "1024 1024 4608 TN" "1024 1024 2304 TN" "1024 1024 9216 TN" "576 1024 4096 NT" "288 1024 4096 NT" "2304 1024 1024 NT"; do
  echo -n "$shape default: "; env -u VILCO_GEMM_BM -u VILCO_GEMM_KS python tools/gemm_one.py f16x2 $shape 2>/dev/null | tail -1
  for ks in 2 3 4 5 6 7 8; do
    echo -n "$shape KS=$ks: "; VILCO_GEMM_KS=$ks python tools/gemm_one.py f16x2 $shape 2>/dev/null | tail -1
  done
done
