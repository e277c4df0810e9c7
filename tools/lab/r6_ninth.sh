mkdir -p gpurun_out
bash tools/lab/dp2_one_gpu.sh 2>&1 | tail -12
python - <<'PY'
import json
for seg in (1, 0):
    try:
        d = json.loads(open('gpurun_out/dp2_seg%d.json' % seg).read().strip().split('\n')[-1])
        print(seg, d['n_gpus'], d['ms_per_step'], d['multi_gpu']['step_mode'][:120])
    except Exception as e:
        print(seg, 'no line', e)
PY
