mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_qkvpre_gpu.py tests/test_dist_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
bash tools/ab_bench.sh "VILCO_PACK_GROUP=0" "VILCO_PACK_GROUP=1" "VILCO_PACK_GROUP=0" "VILCO_PACK_GROUP=1" 2>&1 | tee gpurun_out/r06_ab_pack_group.txt
