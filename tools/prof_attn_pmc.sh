# dynamic instruction mix of the attention kernels (per wave): VALU / SALU / LDS / MFMA instruction counters
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES --output-format csv -d /tmp/pq -o q -- python3 $R/tools/attn_bench.py f16x2 > /tmp/pq_out.txt 2>&1
grep -v "amdgpu\|rocprofv3\|^W2\|^E2" /tmp/pq_out.txt
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/pq/q_counter_collection.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:48]
    key = (n, r['Grid_Size'])
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': cnt[key] += 1
for key, c in agg.items():
    if 'attn' not in key[0]: continue
    w = c['SQ_WAVES'] or 1
    print("%-50s grid %-9s launches %2d waves/launch %6d | per wave: VALU %7.0f SALU %6.0f LDS %5.0f MFMA %5.0f" %
          (key[0], key[1], cnt[key], w / cnt[key], c['SQ_INSTS_VALU'] / w, c['SQ_INSTS_SALU'] / w, c['SQ_INSTS_LDS'] / w, c['SQ_INSTS_MFMA'] / w))
PY
