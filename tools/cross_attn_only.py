"""the cross-attention block of bench_targets (T' = 1152, L = 77, D = 1024, H = 16), forward only, launched eagerly: the
workload of tools/prof_targets_mfma.sh (hardware MFMA-busy counters per kernel)"""
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
with torch.no_grad():
    for _ in range(12):
        mha.forward_tm(x, lens, enc, elens)
torch.cuda.synchronize()
