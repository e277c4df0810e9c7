"""the cross-attention block of bench_targets (T' = 1152, L = 77, D = 1024, H = 16), launched eagerly: the workload of
tools/prof_targets_mfma.sh (hardware MFMA-busy counters per kernel).  argv: B [fwd | fwdbwd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vilco_amd.modeling as vm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
mha = vm.MaskedMHA(1024, 16).to(dev)
x, enc = torch.randn(B, 1152, 1024, device=dev), torch.randn(B, 77, 1024, device=dev)
lens = torch.full((B,), 1152, dtype=torch.int32, device=dev)
elens = torch.full((B,), 77, dtype=torch.int32, device=dev)
if len(sys.argv) > 2 and sys.argv[2] == "fwdbwd":
    x.requires_grad_(True); enc.requires_grad_(True)
    for _ in range(12):
        out = mha.forward_tm(x, lens, enc, elens)
        out = out[0] if isinstance(out, tuple) else out
        out.backward(torch.ones_like(out))
        for p_ in mha.parameters():
            p_.grad = None
        x.grad = None; enc.grad = None
else:
    with torch.no_grad():
        for _ in range(12):
            mha.forward_tm(x, lens, enc, elens)
torch.cuda.synchronize()
