# kernel-stats profile of the default bench.py run (29 steps in the trace: 5 warm-up + 20 timed + 3 profile + 1 optimizer)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 --precision ${PREC:-f16x2} > $R/gpurun_out/${TAG:-step}_prof.log 2>&1
cp /tmp/ps/s_kernel_stats.csv $R/gpurun_out/${TAG:-step}_kernel_stats.csv
python3 $R/tools/step_table.py /tmp/ps/s_kernel_stats.csv 29 24
