#!/bin/bash
# where a tiny GEMM launch spends its microseconds: in-kernel stamps (lab build) + kernel / reduce durations under rocprofv3
mkdir -p gpurun_out
out=gpurun_out/r06_smallgemm.txt; : > $out
for shp in "288 1024 1024 NT" "288 1024 1024 NN" "576 1024 1024 NT" "154 1024 1024 NT" "1152 1024 1024 NT"; do
  echo "== $shp" >> $out
  python tools/lab/gl_stamps.py $shp 2>&1 | grep "group\|rc" >> $out
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sgp -o sg -- python3 $GRAFT_REPO_ROOT/tools/lab/fwd_sweep.py 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - >> $out <<'P'
import csv, glob
f = glob.glob("/tmp/sgp/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print("%-90s calls %6s avg %8.2f us min %8.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
P
cat $out
