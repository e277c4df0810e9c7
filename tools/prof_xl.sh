R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/px
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px -o x -- python3 $R/tools/xlnet_bench.py > /tmp/px_out.txt 2>&1
grep "rel_attention" /tmp/px_out.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/px/x_kernel_stats.csv')))
for r in rows[:16]:
    n=r['Name'].replace('(anonymous namespace)::','')[:64]
    print("  %-64s calls %3s avg %8.1f us total %8.2f ms" % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
