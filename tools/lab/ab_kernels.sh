# per-kernel average durations of a bench step under two library builds (tools/lab/libvilco_prev.so vs current), same box
R=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in prev cur; do
  rm -rf /tmp/pk_$lib
  if [ $lib = prev ]; then export VILCO_HIP_LIB=$R/tools/lab/libvilco_prev.so; else unset VILCO_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$lib -o s -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 --steps 10 --warmup 3 > /tmp/pk_$lib.json 2>/dev/null
done
python3 - <<'PY'
import csv, json
def load(lib):
    rows = list(csv.DictReader(open('/tmp/pk_%s/s_kernel_stats.csv' % lib)))
    return {r['Name'].replace('(anonymous namespace)::', '')[:60]: (int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3) for r in rows}
a, b = load('prev'), load('cur')
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[1] + b.get(k, (0, 0, 0))[1]))
ta = sum(v[1] for v in a.values()); tb = sum(v[1] for v in b.values())
print("total kernel ms: prev %.1f  cur %.1f" % (ta, tb))
for k in keys[:40]:
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    if abs(x[1] - y[1]) > 0.02 * max(x[1], y[1], 1e-9) or 'gemm' in k:
        print("%-60s calls %5d/%5d  avg us %8.1f -> %8.1f   total ms %7.2f -> %7.2f" % (k, x[0], y[0], x[2], y[2], x[1], y[1]))
for lib in ('prev', 'cur'):
    d = json.loads(open('/tmp/pk_%s.json' % lib).read().strip().splitlines()[-1]); print(lib, 'ms/step (profiled)', round(d['ms_per_step'], 2))
PY
