R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg2
rocprofv3 --kernel-trace --output-format csv -d /tmp/pg2 -o g -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 5 > /dev/null 2>&1
python3 $R/tools/gaps.py /tmp/pg2/g_kernel_trace.csv
