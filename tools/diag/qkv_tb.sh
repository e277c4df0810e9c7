# forward-kernel tokens-per-wave sweep (VILCO_QKV_TB) on the T2 target shapes
for tb in 1 2 4; do VILCO_QKV_TB=$tb python tools/bench_targets.py 2>/dev/null | python tools/diag/print_targets.py TB=$tb; done
