# round 6, first GPU call: the reproducer matrix, the changed tests, a baseline bench line
mkdir -p gpurun_out
( for s in 0 1 2; do for p in 0 1 2 3; do timeout 120 tools/lab/nullstream_repro $s $p 3000 20000; done; done
  timeout 120 tools/lab/nullstream_repro 0 3 20000 2000; timeout 120 tools/lab/nullstream_repro 0 2 20000 2000 ) > gpurun_out/r06_nullstream_repro.txt 2>&1
tail -15 gpurun_out/r06_nullstream_repro.txt
timeout 1500 python -m pytest tests/test_nms_gpu.py tests/test_graph_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06_first_tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-targets --extra-batch 0 > gpurun_out/r06_first_bench.json 2> gpurun_out/r06_first_bench.err; cut -c1-400 gpurun_out/r06_first_bench.json
