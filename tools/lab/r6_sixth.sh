mkdir -p gpurun_out
( echo "== null stream, collective replaced by its stream / event choreography (torch streams, no RCCL call)"
  VILCO_GRAPH_OWN_STREAM=0 VILCO_DP_DEBUG_NO_COLLECTIVE=2 TWICE=0 CALLS=3,12,25,40 timeout 600 python tools/lab/dp_staged_dbg2.py 2>&1 | grep "^call\|next"
  echo "== null stream, RCCL, TORCH_NCCL_AVOID_RECORD_STREAMS=1"
  TORCH_NCCL_AVOID_RECORD_STREAMS=1 VILCO_GRAPH_OWN_STREAM=0 TWICE=0 CALLS=3,12,25,40 timeout 600 python tools/lab/dp_staged_dbg2.py 2>&1 | grep "^call\|next"
  echo "== null stream, RCCL, one graph + exchange after it (segments off)"
  VILCO_DP_SEGMENTS=0 VILCO_DP_REPLAY_SYNC=0 VILCO_GRAPH_OWN_STREAM=0 TWICE=0 CALLS=3,12,25,40 timeout 600 python tools/lab/dp_staged_dbg2.py 2>&1 | grep "^call\|next"
) > gpurun_out/r06_nullstream_bisect2.txt 2>&1
cut -c1-260 gpurun_out/r06_nullstream_bisect2.txt
bash tools/lab/instep_gap.sh > gpurun_out/r06_instep_gap.txt 2>&1; cat gpurun_out/r06_instep_gap.txt
