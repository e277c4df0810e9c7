# HBM traffic of the target kernels (tools/bench_targets.py): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes,
# per-kernel sums -> gpurun_out/${TAG}_targets_pmc_{fetch,write}.json  (FETCH_SIZE is doubled on gfx950 when read: see
# tools/finalize_profiles.py / MI355X_MICROARCH.md)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptf /tmp/ptw
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/ptf -o f -- python3 $R/tools/bench_targets.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/ptw -o w -- python3 $R/tools/bench_targets.py > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/ptf/f_counter_collection.csv $R/gpurun_out/${TAG:-r03_z}_targets_pmc_fetch.json | grep -i qkv
python3 $R/tools/pmc_summary.py /tmp/ptw/w_counter_collection.csv $R/gpurun_out/${TAG:-r03_z}_targets_pmc_write.json | grep -i qkv
