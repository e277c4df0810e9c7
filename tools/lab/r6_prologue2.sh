#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r06_prologue_ab.txt; : > $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or linear or conv" 2>&1 | tail -1 >> $out
VILCO_GEMM_SKINNY=0 python tools/lab/gl_stamps.py 4608 1024 1024 NT 2>&1 | grep "group\|prologue" >> $out
N=5 bash tools/lab/ab_lib.sh >> $out 2>&1
cat $out
