"""host floating-point environment before / after the tests that change the CPU oracle's results (see state_dbg.py)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
libm = ctypes.CDLL("libm.so.6")
def env(tag):
    buf = (ctypes.c_ubyte * 32)()
    libm.fegetenv(buf)
    mxcsr = int.from_bytes(bytes(buf[28:32]), 'little'); x87cw = int.from_bytes(bytes(buf[0:2]), 'little')
    a = torch.randn(1 << 16, generator=torch.Generator().manual_seed(3))
    print(tag, "fegetround", libm.fegetround(), "mxcsr 0x%04x" % mxcsr, "x87cw 0x%04x" % x87cw, "threads", torch.get_num_threads(),
          "affinity", len(os.sched_getaffinity(0)), "sum probe %.9e" % float((a * 1e-3).sum()), flush=True)
env("fresh")
import pytest
for k in ("reducer_paths", "run_episodes_end_to_end and True"):
    pytest.main([os.path.join(ROOT, "tests/test_dist_gpu.py"), os.path.join(ROOT, "tests/test_episode.py"), "-q", "-m", "gpu", "-k", k, "-p", "no:cacheprovider"])
    env("after " + k)
