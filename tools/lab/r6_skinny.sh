#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "few_row or test_linear or gemm" 2>&1 | tail -5 > gpurun_out/r06_skinny.txt
timeout 300 python tools/lab/skinny_ab.py 2>&1 | grep -v amdgpu >> gpurun_out/r06_skinny.txt
cat gpurun_out/r06_skinny.txt
