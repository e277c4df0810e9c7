#!/bin/bash
# the two sub-round plan rules (VILCO_GEMM_SUB192): sweep on the new plan, GEMM tests, then same-box A/B of the step
python tools/lab/fwd_sweep.py 2>&1 | grep -v amdgpu | grep "1152 x  4096\|2304 x  1024 x  4096" > gpurun_out/r06_sub192.txt
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or linear or conv" 2>&1 | tail -2 >> gpurun_out/r06_sub192.txt
bash tools/ab_bench.sh "VILCO_GEMM_SUB192=0" "VILCO_GEMM_SUB192=1" "VILCO_GEMM_SUB192=0" "VILCO_GEMM_SUB192=1" >> gpurun_out/r06_sub192.txt 2>&1
cat gpurun_out/r06_sub192.txt
