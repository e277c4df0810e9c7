#!/bin/bash
# does a GEMM of the step take more CYCLES in the step than alone (cache state), or the same cycles at a lower clock (DVFS)?
# GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) per dispatch: (a) inside a P step, (b) the same shapes launched back to back
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc1 /tmp/pc2
VILCO_BENCH_SETTLE_S=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pc1 -o a -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pc2 -o b -- python3 $R/tools/cross_attn_only.py 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
for tag, f in (("in the P step", "/tmp/pc1/a_counter_collection.csv"), ("back to back (cross_attn_only B=2: 2304x1024x1024)", "/tmp/pc2/b_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and 'gemm_pp_kernel' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].replace('(anonymous namespace)::', '')[:52], r['Grid_Size'])].append(float(r['Counter_Value']) / 8)
    print("==", tag)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
        v.sort()
        print("%-54s grid %-8s x%4d  cycles min %8.0f med %8.0f max %8.0f" % (k[0], k[1], len(v), v[0], v[len(v) // 2], v[-1]))
PY
