mkdir -p gpurun_out
# (1) tall single-part dW plan: same-box A/B of the whole step, twice each
bash tools/ab_bench.sh "VILCO_GEMM_DW_TALL=0" "VILCO_GEMM_DW_TALL=1" "VILCO_GEMM_DW_TALL=0" "VILCO_GEMM_DW_TALL=1" 2>&1 | tee gpurun_out/r06_ab_dw_tall.txt
# (2) the null-stream hazard at HEAD: staged replay forced onto the null stream, one-rank RCCL, with and without the collective
( VILCO_GRAPH_OWN_STREAM=0 TWICE=0 CALLS=3,12,25,40 timeout 600 python tools/lab/dp_staged_dbg2.py
  VILCO_GRAPH_OWN_STREAM=0 VILCO_DP_DEBUG_NO_COLLECTIVE=1 TWICE=0 CALLS=3,12,25,40 timeout 600 python tools/lab/dp_staged_dbg2.py
  VILCO_GRAPH_OWN_STREAM=1 TWICE=0 CALLS=3,12,25,40 timeout 600 python tools/lab/dp_staged_dbg2.py ) > gpurun_out/r06_nullstream_bisect.txt 2>&1
grep -v "^\[\|Warning\|warn" gpurun_out/r06_nullstream_bisect.txt | tail -30
# (3) cross-attention block: hardware MFMA-busy counters at B = 8, 16, 32, forward and forward + backward
for B in 8 16 32; do for M in fwd fwdbwd; do TAG=r06 bash tools/prof_targets_mfma.sh $B $M > gpurun_out/r06_cross_attn_mfma_B${B}_$M.txt 2>&1; tail -1 gpurun_out/r06_cross_attn_mfma_B${B}_$M.txt; done; done
# (4) in-step vs back-to-back GEMM counters
bash tools/lab/instep_gap.sh > gpurun_out/r06_instep_gap.txt 2>&1; cat gpurun_out/r06_instep_gap.txt
# (5) what ATen still launches in a step
timeout 600 python tools/diag/aten_sites.py > gpurun_out/r06_aten_sites.txt 2>&1; grep " x " gpurun_out/r06_aten_sites.txt | head -40
