import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
print(json.dumps(bench.side_config("W", torch.device("cuda:0"))))
