# Hardware counters of the MFMA GEMM kernels per shape (round 5): rocprofv3 PMC passes over
# tools/gemm_one.py (operands packed once, 30 back-to-back launches):
#   pass 1  SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy cycles, summed over the 1024 SIMDs) / GRBM_GUI_ACTIVE (elapsed cycles,
#           summed over the 8 XCDs):  busy = BUSY / (GUI_ACTIVE / 8 * 1024)
#   pass 2  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (conflict share of the LDS-array cycles), SQ_INSTS_LDS, SQ_INSTS_VALU_MFMA_MOPS... per launch
# Counters only, no tracing domains.   SHAPES="M N K FORM[ PREC];..." bash tools/prof_gemm_mfma.sh
R=$PWD
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra SH <<< "${SHAPES:-9082 1024 3072 NT;4608 4096 1024 NT;4608 1024 1024 NT;4608 1024 1024 NN;4608 1024 4096 NT;8192 8192 8192 NT;1024 4096 4608 TN 4;1024 1024 4608 TN 4}"
for shp in "${SH[@]}"; do
  rm -rf /tmp/pm1 /tmp/pm2
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm1 -o m -- python3 $R/tools/gemm_one.py f16x2 $shp > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pm2 -o m -- python3 $R/tools/gemm_one.py f16x2 $shp > /dev/null 2>&1
  SHP="$shp" python3 - <<'PY'
import csv, collections, os
def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    first = None
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        if 'gemm_gl_' not in n and 'gemm_pp_kernel' not in n and 'gemm_sp_kernel' not in n: continue
        n = n[:48]
        if first is None: first = r['Counter_Name']
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == first: cnt[n] += 1
    return agg, cnt
a1, c1 = load('/tmp/pm1/m_counter_collection.csv')
a2, c2 = load('/tmp/pm2/m_counter_collection.csv')
print("== %s" % os.environ['SHP'])
for n, c in a1.items():
    g = c['GRBM_GUI_ACTIVE'] / 8.0
    line = "  %-48s x%3d  %9.0f cycles/launch  MFMA busy %5.1f %%" % (n, c1[n], g / c1[n], 100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 1024))
    d = a2.get(n)
    if d and d['SQ_LDS_IDX_ACTIVE'] > 0:
        line += "  | LDS conflict %4.1f %% of LDS cycles, %5.0f LDS insts + %5.0f VALU(+MFMA) insts per wave-launch-unit, waiting %4.1f %% of wave cycles" % (
            100 * d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE'], d['SQ_INSTS_LDS'] / c2[n] / 1e3, d['SQ_INSTS_VALU'] / c2[n] / 1e3,
            100 * d['SQ_WAIT_ANY'] / max(d['SQ_WAVE_CYCLES'], 1.0))
    print(line)
PY
done
