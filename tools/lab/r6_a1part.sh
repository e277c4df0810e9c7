#!/bin/bash
# the full-size parity comparison with every activation / gradient operand of the GEMMs as ONE fp16 part, the weights keeping two
mkdir -p gpurun_out
export VILCO_PRODUCER_PLANES=0 VILCO_LN_PLANES=0 VILCO_ATTN_PLANES=0 VILCO_CONV_DZ_PLANES=0
VILCO_DIAG_OUT=r06_a1part_decisions.json VILCO_LAB_A1PART=1 timeout 1500 python tools/diag/p_parity_decisions.py 2 > gpurun_out/r06_a1part_parity.txt 2>&1
cut -c1-400 gpurun_out/r06_a1part_parity.txt | grep -v "ReLU sign events" | tail -24
