#!/bin/bash
timeout 300 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "pack or gemm or linear" 2>&1 | tail -2
for cap in 2048 1024 768 512 256; do VILCO_PACK_CAP=$cap python3 tools/lab/pack_time.py 2>/dev/null | head -4; done
