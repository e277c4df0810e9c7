mkdir -p gpurun_out
run() { name=$1; shift; timeout 900 python -X faulthandler -m pytest tests/test_dist_gpu.py tests/test_graph_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r06_crash2_$name.txt 2>&1; echo "== $name rc=$? : $(grep -m1 -n 'passed\|failed\|Fatal' gpurun_out/r06_crash2_$name.txt | cut -c1-120) | $(grep -m1 'File \"/root/repo/tests' gpurun_out/r06_crash2_$name.txt | cut -c1-120)"; }
run default
run default_again
VILCO_GRAPH_SHARED_STREAM=1 run shared
VILCO_GRAPH_OWN_STREAM=0 run nullstream
HIP_LAUNCH_BLOCKING=0 AMD_LOG_LEVEL=0 run default3
