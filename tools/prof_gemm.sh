# per-kernel time of one GEMM call (amax / pack / mfma / reduce) for a few shapes; SHAPES="M N K FORM;..."
R=$PWD
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra SH <<< "${SHAPES:-4608 1024 1024 NT;4608 4096 1024 NT;4608 1024 4096 NT;154 1024 1024 NT}"
for shp in "${SH[@]}"; do
  rm -rf /tmp/pg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o g -- python3 $R/tools/gemm_one.py ${PREC:-f16x2} $shp > /dev/null 2>&1
  echo "== $shp"
  python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/pg/g_kernel_stats.csv')))
for r in rows:
    n=r['Name'].replace('(anonymous namespace)::','')[:60]
    if 'randn' in n or 'distribution' in n: continue
    print("  %-60s calls %3s avg %8.1f us" % (n, r['Calls'], float(r['AverageNs'])/1e3))
PY
done
