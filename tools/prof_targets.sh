# kernel-level durations of tools/bench_targets.py (rocprofv3 kernel trace): the fused q/k/v pre-projection and the kernels of the cross-attention block
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python3 $R/tools/bench_targets.py > $R/gpurun_out/${TAG:-targets}.json 2> $R/gpurun_out/${TAG:-targets}.err
cp /tmp/pt/t_kernel_stats.csv $R/gpurun_out/${TAG:-targets}_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open('/tmp/pt/t_kernel_stats.csv')))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:25]:
    print("%-110s calls %5d avg %9.1f us min %9.1f max %9.1f" % (r['Name'][:110], int(r['Calls']), float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
