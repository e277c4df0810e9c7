R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o a -- python3 $R/tools/attn_bench.py f16x2 > /tmp/pa_out.txt 2>&1
grep -v amdgpu /tmp/pa_out.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/pa/a_kernel_stats.csv')))
for r in rows[:10]:
    n=r['Name'].replace('(anonymous namespace)::','')[:60]
    print("  %-60s calls %3s avg %8.1f us" % (n, r['Calls'], float(r['AverageNs'])/1e3))
PY
