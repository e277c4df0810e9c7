# which HIP stream carries what in the replayed step (kernel trace of a short bench.py run -> tools/step_streams.py)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
VILCO_BENCH_SETTLE_S=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o s -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 --steps 10 --warmup 3 > $R/gpurun_out/streams_prof.log 2>&1
python3 $R/tools/step_streams.py /tmp/ps/s_kernel_trace.csv > $R/gpurun_out/step_streams.txt 2>&1
python3 $R/tools/step_gaps.py /tmp/ps/s_kernel_trace.csv > $R/gpurun_out/step_gaps.txt 2>&1
