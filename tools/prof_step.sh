# kernel-stats profile of a short bench.py run; the step count of the trace is taken from the GEMM launch count (GEMMS_PER_STEP, default 321 at config P)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
VILCO_BENCH_SETTLE_S=${SETTLE:-1} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 --steps ${STEPS:-10} --warmup 3 --precision ${PREC:-f16x2} > $R/gpurun_out/${TAG:-step}_prof.log 2>&1
cp /tmp/ps/s_kernel_stats.csv $R/gpurun_out/${TAG:-step}_kernel_stats.csv
NS=$(python3 - <<'PY'
import csv
n = sum(int(r['Calls']) for r in csv.DictReader(open('/tmp/ps/s_kernel_stats.csv')) if 'gemm_pp_kernel' in r['Name'] or 'gemm_gl_' in r['Name'] or 'gemm_sp_kernel' in r['Name'])
import os
print(max(1, round(n / int(os.environ.get('GEMMS_PER_STEP', '321')))))
PY
)
python3 $R/tools/step_table.py /tmp/ps/s_kernel_stats.csv $NS ${ROWS:-24}
