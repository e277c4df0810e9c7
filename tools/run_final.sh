# round-end measurement bundle: bench line (with cpu_baseline, targets, side configs), rocprofv3 kernel stats of the SAME
# workload (bench.py --no-cpu-baseline --no-targets --extra-batch 0), PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately)
R=$PWD; TAG=${TAG:-r03_z}
if [ -z "$SKIP_BENCH" ]; then python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300; fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 > $R/gpurun_out/${TAG}_prof_bench.json 2> /dev/null
cp /tmp/ps/s_kernel_stats.csv $R/gpurun_out/${TAG}_bench_kernel_stats.csv
VILCO_BENCH_SETTLE_S=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 --steps 2 --warmup 1 > /dev/null 2>&1
VILCO_BENCH_SETTLE_S=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python3 $R/bench.py --no-cpu-baseline --no-targets --extra-batch 0 --steps 2 --warmup 1 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/pf/f_counter_collection.csv $R/gpurun_out/${TAG}_pmc_fetch.json > /dev/null
python3 $R/tools/pmc_summary.py /tmp/pw/w_counter_collection.csv $R/gpurun_out/${TAG}_pmc_write.json > /dev/null
# extras cited in DESIGN.md: per-shape GEMM table, target-kernel kernel stats, attention instruction mix
cd $R
python3 tools/gemm_shapes.py 3 > gpurun_out/${TAG}_gemm_shapes.txt 2>/dev/null
TAG=${TAG}_targets bash tools/prof_targets.sh > /dev/null 2>&1
bash tools/prof_attn_pmc.sh > gpurun_out/${TAG}_attn_insts.txt 2>&1
ls -la $R/gpurun_out | grep ${TAG}
TAG=${TAG} bash tools/prof_targets_pmc.sh > gpurun_out/${TAG}_targets_pmc.txt 2>&1
# round 5: matrix-pipe busy / LDS counters of the GEMM kernels per shape
cd $R
TAG=${TAG} bash tools/prof_gemm_mfma.sh > gpurun_out/${TAG}_gemm_mfma.txt 2>&1
