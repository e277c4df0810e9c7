"""Golden vectors of the EPISODE path from the IMPORTED REFERENCE (this container only).
Run:  python tests/golden/make_golden_episode.py   ->  tests/golden/episode_vilco.pt

The reference's own functions are driven exactly as MQ/train_cl.py drives them: make_optimizer / make_scheduler,
`train_one_epoch` (train_utils.py:278-423; LOCAL_RANK=0, a list of batches as the loader), `post_train_step` inside
it, eval-mode forward with the adapter-EMA ensemble (meta_archs.py:854-881), add_samples_to_mem, n_known,
`augment_classification`, a NEW optimizer and scheduler, and the second task with task_id = 1 (so the L2P pull term
of meta_archs.py:1478-1480 is live).  Recorded: every iteration's loss dict and LR, the state after each task
(large adapter tensors as strided samples, cases.compact), the EMA adapters, the post-augment head tensors, and the
eval outputs (raw ensemble logits / offsets of clip 0 and the decoded + NMS'd segments)."""
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402


class _Log:
    def info(self, *a, **k):
        pass


def snapshot(model):
    return {k: cases.compact(v) for k, v in model.state_dict().items()}


def main():
    os.environ["LOCAL_RANK"] = "0"
    libs = ref_import.setup(extra_xlnet=((32, 4), (64, 4), (128, 4), (2304, 16)))
    from libs.modeling import make_meta_arch
    from libs.utils import make_optimizer, make_scheduler, train_one_epoch
    over = cases.episode_overrides()
    cfg = ref_import.make_cfg(libs, **over)
    torch.manual_seed(4321)
    model = make_meta_arch(cfg['model_name'], **cfg['model'])
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if p_.numel() <= cases.BIG and (p_.dim() <= 1 or 'norm' in n_ or n_.startswith(('mu', 'sigma'))):
                p_.add_(0.05 * torch.randn(p_.shape, generator=g))
    cases.perturb_episode_state(model)
    for ema in model.pets_emas:                      # the EMA copy was taken at construction: restart it from the new state
        ema.set(model.pets)
    out = {'overrides': over, 'init_state': snapshot(model), 'tasks': []}

    seen = []
    orig_forward = model.forward

    def recording_forward(*a, **k):
        r = orig_forward(*a, **k)
        if isinstance(r, dict) and 'final_loss' in r:
            seen.append({kk: float(v) for kk, v in r.items()})
        return r
    model.forward = recording_forward

    optimizer = make_optimizer(model, cfg['opt'])
    scheduler = make_scheduler(optimizer, cfg['opt'], len(cases.episode_batches(0)))
    for task in range(2):
        batches = cases.episode_batches(task)
        seen.clear()
        lrs = []
        orig_step = scheduler.step

        def step_rec(*a, **k):
            lrs.append(optimizer.param_groups[0]['lr'])
            return orig_step(*a, **k)
        scheduler.step = step_rec
        model.pre_train_epoch(task_id=task, current_epoch=0)
        train_one_epoch(batches, model, optimizer, scheduler, 0, 1, model_ema=None,
                        clip_grad_l2norm=cfg['train_cfg']['clip_grad_l2norm'], print_freq=1000, logger=_Log(),
                        cl_name=cfg['cl_cfg']['name'], reg_lambda=cfg['cl_cfg']['reg_lambda'],
                        prev_out_cls_logits_dict={}, current_task_id=task)
        rec = {'losses': list(seen), 'lrs': list(lrs), 'loss_normalizer': float(model.loss_normalizer),
               'state': snapshot(model)}
        model.eval()
        with torch.no_grad():
            clip = cases.episode_batches(task)[0][0]
            raw = model([clip], task_id=task, is_training=False, get_emb=True)
            res = model([clip], task_id=task, is_training=False)[0]
        rec['eval_cls_logits'] = [x.clone() for x in raw[0]]
        rec['eval_offsets'] = [x.clone() for x in raw[1]]
        rec['inference'] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in res.items()}
        # replay memory + class-head growth, as train_cl.py:343-389
        n_cls = model.cls_head.cls_head.conv.out_channels
        data = {}
        for b in batches:
            for v in b:
                for c in v['labels'].tolist():
                    if (task == 0 and c < cases.EP_NCLS0) or (task == 1 and c >= cases.EP_NCLS0):
                        data.setdefault(c, []).append(v)
        random.seed(0)
        model.add_samples_to_mem(None, data, cfg['cl_cfg']['memory_size'] // n_cls)
        model.n_known = len(model.memory)
        rec['n_known'] = model.n_known
        rec['memory_ids'] = {c: [v['video_id'] for v in vs] for c, vs in model.memory.items()}
        if task == 0:
            torch.manual_seed(99)                     # the new head rows are randomly initialised (blocks.py:85-104)
            model.augment_classification(cases.EP_NEW, 'cpu')
            rec['post_augment'] = {k: v.clone() for k, v in model.state_dict().items()
                                   if k.startswith(('cls_head.cls_head', 'mu', 'sigma'))}
            optimizer = make_optimizer(model, cfg['opt'])
            scheduler = make_scheduler(optimizer, cfg['opt'], len(cases.episode_batches(0)))
        out['tasks'].append(rec)
        print('task', task, [round(l['final_loss'], 5) for l in rec['losses']], 'lrs', rec['lrs'], 'n_known', rec['n_known'],
              'segs', tuple(res['segments'].shape))
    path = os.path.join(HERE, 'episode_vilco.pt')
    torch.save(out, path)
    print('%.1f KB' % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
