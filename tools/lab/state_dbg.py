"""does the HIP P step depend on what ran earlier in the process?  `python state_dbg.py fresh|hist` -> gpurun_out/state_<mode>.pt"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mode = sys.argv[1]
if mode == "hist":
    import pytest
    pytest.main([os.path.join(ROOT, "tests/test_dist_gpu.py"), os.path.join(ROOT, "tests/test_episode.py"), "-q", "-m", "gpu", "-k",
                 "reducer_paths or (run_episodes_end_to_end and True)", "-p", "no:cacheprovider"])
import bench
import vilco_amd.modeling as vm
from vilco_amd import ops, _lib
from vilco_amd.modeling import blocks
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
batch = bench.synth_batch(2, dev)
torch.manual_seed(5); torch.cuda.manual_seed_all(5)
ops._drop_counter[0] = 0
blocks.reset_drop_pool()
_lib.check(_lib.load().vilco_seed_word_set(0, None))
ops.dropout_log = []
losses = model(batch, is_training=True)
losses['final_loss'].backward()
torch.cuda.synchronize()
log = [(e[0], float(e[1].sum()) if e[0] == "droppath" else e[1:]) for e in ops.dropout_log]
ops.dropout_log = None
out = {"losses": {k: float(v) for k, v in losses.items()}, "log": log,
       "grads": {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}}
torch.save(out, os.path.join(ROOT, "gpurun_out", "state_%s.pt" % mode))
print(mode, out["losses"], len(log))
