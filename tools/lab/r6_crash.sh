mkdir -p gpurun_out
run() { name=$1; shift; timeout 1500 python -X faulthandler -m pytest "$@" -x -q -m gpu -p no:cacheprovider > gpurun_out/r06_crash_$name.txt 2>&1; echo "== $name rc=$? : $(grep -m1 -n 'passed\|failed\|Fatal' gpurun_out/r06_crash_$name.txt | cut -c1-120) | $(head -1 gpurun_out/r06_crash_$name.txt | cut -c1-80)"; }
run B tests/test_dist_gpu.py tests/test_graph_gpu.py
run D tests/test_episode.py tests/test_graph_gpu.py
run C tests/test_fullsize_gpu.py tests/test_graph_gpu.py -k "not mask_realisations"
run E tests/test_cl_parts.py tests/test_distill.py tests/test_eval_formats.py tests/test_graph_gpu.py
run A tests/test_cl_parts.py tests/test_dist_gpu.py tests/test_distill.py tests/test_episode.py tests/test_eval_formats.py tests/test_fullsize_gpu.py tests/test_graph_gpu.py -k "not mask_realisations"
