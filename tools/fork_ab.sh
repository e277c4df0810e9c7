#!/bin/bash
# which chains to fork onto side streams inside the captured step: same-box A/B of the replayed P step (interleaved repeats)
for rep in 1 2; do
for f in "" text heads text,heads; do
  echo -n "VILCO_GRAPH_STREAMS='$f'  "
  VILCO_GRAPH_STREAMS="$f" PROBE_SKIP_EAGER=1 PROBE_FB_ONLY=1 timeout 300 python tools/graph_probe.py P 25 2>&1 | grep -E "graph fwd|rror"
done
done
