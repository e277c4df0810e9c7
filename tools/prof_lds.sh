R=$PWD
cd /tmp && export TMPDIR=/tmp
for shp in "4096 1024 4608 TN" "4608 1024 1024 NN" "4608 1024 1024 NT"; do
  rm -rf /tmp/pl
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pl -o l -- python3 $R/tools/gemm_one.py f16x2 $shp > /dev/null 2>&1
  echo "== $shp"; python3 $R/tools/pmc_summary.py /tmp/pl/l_counter_collection.csv /tmp/pl/sum.json | grep -E "gemm_pp|pack" | cut -c1-200
done
