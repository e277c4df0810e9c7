# kernel-stats profile of bench.py (per-step numbers); usage: PREC=f16x2 TAG=r01_f bash tools/prof_step.sh
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --precision ${PREC:-split3} > $R/gpurun_out/${TAG:-step}_prof.log 2>&1
cp /tmp/ps/s_kernel_stats.csv $R/gpurun_out/${TAG:-step}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/ps/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total GPU ms (all steps)', tot/1e6)
for r in rows[:26]:
    n=r['Name'].replace('(anonymous namespace)::','')[:70]
    print(f"{n:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:8.2f}ms {float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}%")
PY
