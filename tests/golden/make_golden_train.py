"""Golden vectors for the step glue (SURVEY.md 8f-1 pins): parameter-group membership produced by the
reference's make_optimizer and LR sequences of its schedulers.  Run: python tests/golden/make_golden_train.py"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402

libs = ref_import.setup(extra_xlnet=((32, 4), (64, 4), (128, 4), (2304, 16)))
from libs.modeling import make_meta_arch  # noqa: E402
from libs.utils import make_optimizer, make_scheduler  # noqa: E402

out = {}
for name in ("xl", "prompt"):
    okw, L = cases.CASES[name]
    over = cases.overrides(**okw)
    if name == "prompt":
        over['dataset']['max_seq_len'] = 1024
        over['cl_cfg'].update(use_adapt=True, adapt_blocks=[0, 1])
    cfg = ref_import.make_cfg(libs, **over)
    model = make_meta_arch(cfg['model_name'], **cfg['model'])
    opt = make_optimizer(model, dict(cfg['opt'], weight_decay=0.05, learning_rate=1e-4))
    by_id = {}
    for n, p in model.named_parameters(remove_duplicate=False):
        by_id.setdefault(id(p), []).append(n)
    groups = [[sorted(by_id[id(p)])[0] for p in g['params']] for g in opt.param_groups]
    out[name] = {"overrides": over, "groups": groups, "weight_decay": [g['weight_decay'] for g in opt.param_groups]}

lin = torch.nn.Linear(2, 2)
seqs = {}
for tag, oc in (("cosine", dict(warmup=True, warmup_epochs=5, epochs=10, schedule_type="cosine", schedule_steps=[], schedule_gamma=0.1)),
                ("multistep", dict(warmup=True, warmup_epochs=2, epochs=8, schedule_type="multistep", schedule_steps=[3, 6], schedule_gamma=0.1))):
    opt = torch.optim.AdamW(lin.parameters(), lr=1e-4)
    sch = make_scheduler(opt, oc, 10)
    lrs = []
    for _ in range((oc['warmup_epochs'] + oc['epochs']) * 10):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        sch.step()
    seqs[tag] = {"cfg": oc, "iters_per_epoch": 10, "base_lr": 1e-4, "lrs": lrs}
out["lr"] = seqs
json.dump(out, open(os.path.join(HERE, "train_glue.json"), "w"))
print({k: [len(g) for g in v["groups"]] for k, v in out.items() if k != "lr"}, seqs["cosine"]["lrs"][:4])
