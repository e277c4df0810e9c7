for i in 1 2; do
for v in 1 0; do
VILCO_OPT_AMAX=$v python bench.py --no-cpu-baseline --no-targets --extra-batch 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d['optimizer_step']; print('OPT_AMAX=$v', round(d['ms_per_step'],2), round(o['ms'],3), round(o['train_iteration_ms_measured'],2))"
done; done
