# forced tile heights on the 75 %-fill shapes (default: the cost model's choice)
for shape in "4608 1024 1024 NT" "4608 1024 1024 NN" "4608 1024 4096 NT" "4608 4096 1024 NT" "2304 1024 1024 NT" "9082 1024 3072 NT"; do
  for bm in default 128 192 256; do
    if [ $bm = default ]; then unset VILCO_GEMM_BM; else export VILCO_GEMM_BM=$bm; fi
    echo -n "$shape BM=$bm: "; python tools/gemm_one.py f16x2 $shape 2>/dev/null | tail -1
  done
done
