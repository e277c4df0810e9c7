"""Same-box A/B of a bench.py side configuration (cfg1 / W) under environment toggles given as arguments:
python tools/lab/side_ab.py cfg1 VILCO_GEMM_GL=0 VILCO_GEMM_GL=1 ...   (each toggle runs in a child process, interleaved twice)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, toggles = sys.argv[1], sys.argv[2:]
code = ("import sys, json; sys.path.insert(0, %r); import torch, bench; r = bench.side_config(%r, torch.device('cuda:0'), steps=20, warm=3); "
        "print('SIDE', json.dumps({k: r[k] for k in ('ms_per_step', 'clips_per_s') if k in r}))" % (ROOT, name))
for rnd in range(2):
    for t in toggles:
        env = dict(os.environ)
        for kv in t.split(","):
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("SIDE")]
        print(name, t, line[0] if line else out.stderr[-400:], flush=True)
