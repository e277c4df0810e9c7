# split-K / tile-height sweep on the K = 4096 shapes
for shape in "4608 1024 4096 NT" "4608 1024 4096 NN" "2304 1024 4096 NT" "1152 1024 4096 NT"; do
  echo -n "$shape default: "; env -u VILCO_GEMM_BM -u VILCO_GEMM_KS python tools/gemm_one.py f16x2 $shape 2>/dev/null | tail -1
  for bm in 128 192 256; do for ks in 1 2 3 4; do
    echo -n "$shape BM=$bm KS=$ks: "; VILCO_GEMM_BM=$bm VILCO_GEMM_KS=$ks python tools/gemm_one.py f16x2 $shape 2>/dev/null | tail -1
  done; done
done
