#!/bin/bash
# fused q/k/v pre-projection: ring kernel (VILCO_QKV_RING=1) vs the three-pass kernel; parity first, then timing
[ -z "$SKIP_TESTS" ] && VILCO_QKV_RING=1 timeout 900 python3 -m pytest tests/test_qkvpre_gpu.py -x -q -m gpu 2>&1 | tail -3
for m in 0 1; do
  VILCO_QKV_RING=$m python3 tools/qkv_ab.py 2>/dev/null | tail -13
done
