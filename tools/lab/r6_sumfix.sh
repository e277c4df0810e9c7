mkdir -p gpurun_out
python tools/lab/sum_graph_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sum_graph_probe.txt; cat gpurun_out/r06_sum_graph_probe.txt
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -p no:cacheprovider -k "replayed_steps_equal" 2>&1 | tail -6
