R=$PWD
python3 tools/gemm_breakdown.py split3 2>&1 | grep -v amdgpu.ids > gpurun_out/r01_e_gemm_breakdown.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r01_e_pmcf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r01_e_pmcw.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmc_fetch/f_counter_collection.csv $R/gpurun_out/r01_e_pmc_fetch.json > /dev/null
python3 $R/tools/pmc_summary.py /tmp/pmc_write/w_counter_collection.csv $R/gpurun_out/r01_e_pmc_write.json > /dev/null
