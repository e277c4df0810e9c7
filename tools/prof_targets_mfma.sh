# hardware MFMA utilisation of the cross-attention block's kernels (north-star target 1): rocprofv3 PMC
# SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy cycles, summed over the 1024 SIMDs) against GRBM_GUI_ACTIVE (elapsed cycles,
# summed over the 8 XCDs) per dispatch:  util = BUSY / (GUI_ACTIVE / 8 * 1024).  Counters only, no tracing domains.
R=$PWD; B=${1:-8}; MODE=${2:-fwd}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -o m -- python3 $R/tools/cross_attn_only.py $B $MODE > /dev/null 2>&1
python3 - <<PY
import csv, collections, json
rows = list(csv.DictReader(open('/tmp/pm/m_counter_collection.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[n] += 1
tb = tg = 0.0; out = []
for n, c in sorted(agg.items(), key=lambda kv: -kv[1]['GRBM_GUI_ACTIVE']):
    g = c['GRBM_GUI_ACTIVE'] / 8.0; b = c['SQ_VALU_MFMA_BUSY_CYCLES']
    if g <= 0: continue
    tb += b; tg += g
    out.append({"kernel": n, "launches": cnt[n], "elapsed_cycles_per_launch": g / cnt[n], "mfma_busy_frac": b / (g * 1024)})
    print("%-62s x%3d  %9.0f cycles/launch  MFMA busy %5.1f %%" % (n, cnt[n], g / cnt[n], 100 * b / (g * 1024)))
print("block (all kernels, cycle-weighted): MFMA busy %.1f %%" % (100 * tb / (tg * 1024)))
json.dump({"B": $B, "mode": "$MODE", "kernels": out, "block_mfma_busy_frac": tb / (tg * 1024)}, open('$R/gpurun_out/${TAG:-r03_z}_cross_attn_mfma_B${B}_$MODE.json', 'w'), indent=1)
PY
