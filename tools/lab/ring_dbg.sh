#!/bin/bash
for d in 0 1 2 3 4 6; do echo "== DBG=$d"; VILCO_QKV_DBG=$d VILCO_QKV_RING=1 python3 tools/qkv_ab.py 2>&1 | tail -4 | head -2|tail -1; done
