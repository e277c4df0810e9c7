#!/bin/bash
# the full-size parity comparison with the stored weights' second fp16 part zeroed (VILCO_LAB_W1PART=1: the 2-MFMA arithmetic, emulated)
mkdir -p gpurun_out
VILCO_DIAG_OUT=r06_w1part_decisions.json VILCO_LAB_W1PART=1 timeout 1500 python tools/diag/p_parity_decisions.py 2 > gpurun_out/r06_w1part_parity.txt 2>&1
cut -c1-400 gpurun_out/r06_w1part_parity.txt | grep -v "ReLU sign events" | tail -40
