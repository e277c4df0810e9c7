#!/bin/bash
# A/B of the split-K finish (in-launch fix-up vs reduce launch) on the same box: step time of cfg1 and P (hipGraph replay)
for fx in 1 0; do
  export VILCO_GEMM_FIXUP=$fx
  echo "== VILCO_GEMM_FIXUP=$fx"
  PROBE_FB_ONLY=1 timeout 200 python tools/graph_probe.py cfg1 50 2>&1 | grep -E "graph fwd|Error|error"
  PROBE_FB_ONLY=1 timeout 200 python tools/graph_probe.py P 30 2>&1 | grep -E "graph fwd|Error|error"
done
