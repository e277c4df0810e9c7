#!/bin/bash
# Round 6: the same GEMM launches INSIDE the P step and BACK TO BACK -- elapsed cycles (GRBM_GUI_ACTIVE / 8),
# L2 hit rate (TCC_HIT / (HIT + MISS)) and bytes requested from the fabric (FETCH_SIZE x 2 on gfx950) per dispatch, grouped by
# (kernel instantiation, grid).  Separate PMC passes, kernel-trace only.  Under PMC every dispatch runs alone (no forked chains).
R=$PWD
cd /tmp && export TMPDIR=/tmp
for pass in GRBM_GUI_ACTIVE "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE; do
  tag=$(echo $pass | cut -c1-7 | tr ' ' '_')
  rm -rf /tmp/ig_s_$tag /tmp/ig_b_$tag
  VILCO_BENCH_SETTLE_S=0 VILCO_BENCH_GRAPH=0 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/ig_s_$tag -o s -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 --steps 2 --warmup 1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/ig_b_$tag -o b -- python3 $R/tools/lab/gemm_b2b.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, collections, glob
def load(prefix):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('/tmp/ig_%s_*/*_counter_collection.csv' % prefix):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
            if 'gemm_gl' not in n and 'gemm_pp' not in n: continue
            out[(n[:58], r['Grid_Size'], r.get('LDS_Block_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
    return out
def med(v):
    v = sorted(v); return v[len(v) // 2] if v else float('nan')
S, B = load('s'), load('b')
print("%-60s %-9s | %-31s | %-31s" % ("kernel (grid)", "", "in the step: n, cycles, L2 hit, MB", "back to back: n, cycles, L2 hit, MB"))
for k in sorted(B, key=lambda k: -med(B[k]['GRBM_GUI_ACTIVE'])):
    if k not in S: continue
    row = []
    for D in (S, B):
        c = D[k]
        h, m = med(c['TCC_HIT_sum']), med(c['TCC_MISS_sum'])
        row.append("%3d %8.0f  %5.1f %%  %7.1f" % (len(c['GRBM_GUI_ACTIVE']), med(c['GRBM_GUI_ACTIVE']) / 8, 100 * h / max(h + m, 1), med(c['FETCH_SIZE']) * 2 / 1024))
    print("%-60s %-9s | %-31s | %-31s" % (k[0], k[1], row[0], row[1]))
PY
