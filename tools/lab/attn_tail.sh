# kernel durations of tools/lab/attn_tail.py by (kernel, grid size)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/at
rocprofv3 --kernel-trace --output-format csv -d /tmp/at -o t -- python3 $R/tools/lab/attn_tail.py $ARGS > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/at/t_kernel_trace.csv')):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'attn' not in n: continue
    agg[(n, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (n, g), v in sorted(agg.items()):
    v = sorted(v)[: max(1, len(v) - 1)]
    print("%-32s workgroups %6d (%.3f rounds of 512)  %8.1f us   per workgroup-round %7.1f us" % (n, g, g / 512, sum(v) / len(v) / 1e3, sum(v) / len(v) / 1e3 / -(-g // 512)))
PY
