R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/r01_e_prof.log 2>&1
cp /tmp/prof_e/e_kernel_stats.csv $R/gpurun_out/r01_e_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r01_e_pmcf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r01_e_pmcw.log 2>&1
ls /tmp/pmc_fetch /tmp/prof_e
python3 $R/tools/pmc_summary.py /tmp/pmc_fetch/f_counter_collection.csv $R/gpurun_out/r01_e_pmc_fetch.json
python3 $R/tools/pmc_summary.py /tmp/pmc_write/w_counter_collection.csv $R/gpurun_out/r01_e_pmc_write.json
