"""Shared definitions of the golden cases: config overrides + seeded synthetic inputs (SURVEY.md 8d
recipe scaled down).  Used by make_golden.py (which runs the imported reference) and by the tests
(which run the oracle / the HIP path on the same inputs)."""
import torch

NCLS = 22


def overrides(D=32, T=64, Cin=48, Ctxt=24, H=4, use_xl=True, droppath=0.1, cl=None, loss_weight=1.0,
              al_loss_weight=0.0):
    o = dict(dataset=dict(input_dim=Cin, num_classes=NCLS, max_seq_len=T),
             model=dict(embd_dim=D, fpn_dim=D, head_dim=D, n_head=H, backbone_arch=(2, 2, 5), use_abs_pe=True,
                        use_cross_modal=True, n_txt_in=Ctxt, max_buffer_len_factor=1.0, use_xl=use_xl),
             train_cfg=dict(init_loss_norm=100, dropout=0.0, droppath=droppath, loss_weight=loss_weight,
                            al_loss_weight=al_loss_weight),
             test_cfg=dict(max_seg_num=200, nms_sigma=0.75, min_score=0.001))
    if cl:
        o['cl_cfg'] = cl
    return o


CASES = {
    # name: (overrides kwargs, text length L)
    "xl": (dict(use_xl=True, droppath=0.1), 9),
    "noxl": (dict(use_xl=False, droppath=0.0, al_loss_weight=0.5), 13),
    "prompt": (dict(use_xl=False, droppath=0.1,
                    cl=dict(name='l2p', prompt_pool=True, pool_size=10, topk=4, length=20, embed_dim=24)), 7),
}


def video_list(T, Cin, Ctxt, L, B=2, seed=0, short=17):
    """seeded clips: feats ~ N(0,1) [Cin, t_b] (t_0 = T, t_1 = T - short), text [Ctxt, L_b]."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for b in range(B):
        t = T if b == 0 else T - short
        lb = L if b == 0 else max(1, L - 2)
        segs = torch.tensor([[2.0, 9.0], [12.5, 30.25]]) if T <= 64 else torch.tensor([[10.0, 40.0], [60.5, 130.25]])
        out.append({'video_id': 'v%d' % b, 'feats': torch.randn(Cin, t, generator=g), 'segments': segs,
                    'labels': torch.tensor([1, 5]), 'fps': 30.0, 'duration': 100.0, 'feat_stride': 16,
                    'feat_num_frames': 16, 'segmentation_labels': torch.zeros(t, NCLS),
                    'prompt_feature': torch.randn(Ctxt, lb, generator=g)})
    return out


# ------------------------------------------------------------------------------------------------------------------
# Episode case (BASELINE configs[2] scaled down): 2 query-incremental tasks x 3 iterations of the ViLCo recipe --
# L2P prompt pool, time adapters (hard-wired to T = 1024 in the reference, meta_archs.py:682) with their EMA,
# class-head growth between the tasks, a new optimizer per task.  Train mode with dropout = droppath = 0 (and the
# tiny XLNet JSON's dropout 0) is deterministic, so the imported reference and the HIP path can be compared step by
# step.  Tensors with more than BIG elements (the adapters' T x 5T Linear layers) are stored as a strided sample.
EP_T, EP_D, EP_CIN, EP_CTXT, EP_H = 1024, 32, 48, 24, 4
EP_NCLS0, EP_NEW = 4, 3
BIG, SAMPLE_STRIDE = 200_000, 997


def episode_overrides():
    o = overrides(D=EP_D, T=EP_T, Cin=EP_CIN, Ctxt=EP_CTXT, H=EP_H, use_xl=True, droppath=0.0, al_loss_weight=0.2,
                  cl=dict(name='l2p', prompt_pool=True, pool_size=10, topk=4, length=5, embed_dim=EP_CTXT,
                          use_adapt=True, adapt_blocks=[0, 1], memory_size=8, type_sampling='random'))
    o['dataset']['num_classes'] = EP_NCLS0
    o['opt'] = dict(type='AdamW', momentum=0.9, weight_decay=0.05, learning_rate=1e-3, epochs=1, warmup=True,
                    warmup_epochs=1, schedule_type='cosine', schedule_steps=[], schedule_gamma=0.1)
    o['train_cfg']['clip_grad_l2norm'] = 1.0
    return o


def episode_clip(idx, labels, T=EP_T, L=9):
    """clip `idx` with two ground-truth moments of the given class ids"""
    g = torch.Generator().manual_seed(5000 + idx)
    t = T - 23 * (idx % 3)
    segs = torch.tensor([[40.0 + 7 * idx, 150.0 + 11 * idx], [300.5 + 5 * idx, 620.25 - 9 * idx]])
    return {'video_id': 'ep%d' % idx, 'id': 'ep%d' % idx, 'feats': torch.randn(EP_CIN, t, generator=g), 'segments': segs,
            'labels': torch.tensor(labels), 'fps': 30.0, 'duration': 400.0, 'feat_stride': 16, 'feat_num_frames': 16,
            'segmentation_labels': torch.zeros(t, 8), 'prompt_feature': torch.randn(EP_CTXT, L - (idx % 2), generator=g)}


def episode_batches(task):
    """four batches of two clips per task; task 1 replays clips of task 0 next to the new classes 4..6"""
    if task == 0:
        return [[episode_clip(0, [0, 1]), episode_clip(1, [2, 3])], [episode_clip(2, [1, 2]), episode_clip(3, [3, 0])],
                [episode_clip(4, [0, 2]), episode_clip(5, [1, 3])], [episode_clip(10, [3, 1]), episode_clip(11, [2, 0])]]
    return [[episode_clip(6, [4, 5]), episode_clip(0, [0, 1])], [episode_clip(7, [6, 4]), episode_clip(8, [5, 6])],
            [episode_clip(9, [4, 6]), episode_clip(2, [1, 2])], [episode_clip(12, [5, 4]), episode_clip(13, [6, 5])]]


def seeded_tensor(name, shape, scale):
    """deterministic stand-in for a tensor too large to store (same bits wherever torch's CPU generator runs)"""
    g = torch.Generator().manual_seed(sum(ord(c) * (i + 1) for i, c in enumerate(name)) % (2 ** 31))
    return scale * torch.randn(shape, generator=g)


def compact(t):
    """what a golden keeps of a tensor: everything, or (for > BIG elements) a strided sample + the two norms"""
    if t.numel() <= BIG:
        return t.clone()
    f = t.detach().reshape(-1).double()
    return {'sample': f[::SAMPLE_STRIDE].float().clone(), 'sum': float(f.sum()), 'l2': float(f.norm()), 'shape': tuple(t.shape)}


def perturb_episode_state(model):
    """shared by the generator (reference model) and the tests (HIP model): the initial weights of the episode case.
    Small tensors are copied from the golden; the adapters' large Linear weights come from `seeded_tensor` (their
    reference init has layer.2 = 0, which would carry no signal)."""
    done = 0
    with torch.no_grad():
        for n, p in model.pets.named_parameters():          # (also reachable as backbone.branch.b.adapters.attn.*)
            if p.numel() > BIG:
                p.copy_(seeded_tensor('pets.' + n, p.shape, 0.02 if 'layer.0' in n else 0.01).to(p.device))
                done += 1
    assert done == 4, done


# ------------------------------------------------------------------------------------------------------------------
# toy module for the EWC / MAS goldens: the regularisers only see named_parameters / reg_params / model(batch)
class RegToy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.body = torch.nn.Linear(6, 5)
        self.head = torch.nn.Linear(5, 7)              # "grown" class head: 7 rows now, 4 when task 0 was consolidated
        self.scale = torch.nn.Parameter(torch.tensor(1.5))          # names containing 'scale' are skipped (EWC.py:15)
        self.unused = torch.nn.Parameter(torch.zeros(3))            # never gets a gradient: absent from the dictionaries
        with torch.no_grad():
            for p in (self.body.weight, self.body.bias, self.head.weight, self.head.bias):
                p.copy_(torch.randn(p.shape, generator=g))
        self.reg_params = {}

    def forward(self, x):
        return {'final_loss': (self.head(torch.tanh(self.body(x))) * self.scale).pow(2).mean()}


def reg_toy_loss(model):
    return model(torch.randn(4, 6, generator=torch.Generator().manual_seed(8)).to(model.scale.device))['final_loss']


def reg_toy_loader(device='cpu'):
    return [torch.randn(4, 6, generator=torch.Generator().manual_seed(20 + i)).to(device) for i in range(3)]


# ------------------------------------------------------------------------------------------------------------------
# NLQ variant (BASELINE configs[3] scaled down): sliding-window attention blocks and the two-stream backbone
NLQ_C, NLQ_H, NLQ_T, NLQ_L, NLQ_WIN, NLQ_CV, NLQ_CT = 32, 4, 96, 11, 9, 40, 24


def nlq_inputs():
    g = torch.Generator().manual_seed(123)
    x = torch.randn(2, NLQ_C, NLQ_T, generator=g)
    mask = (torch.arange(NLQ_T)[None, :] < torch.tensor([NLQ_T, NLQ_T - 21])[:, None]).unsqueeze(1)
    x = x * mask
    txt = torch.randn(2, NLQ_C, NLQ_L, generator=g)
    tmask = (torch.arange(NLQ_L)[None, :] < torch.tensor([NLQ_L, NLQ_L - 3])[:, None]).unsqueeze(1)
    return x, mask, txt * tmask, tmask


def nlq_backbone_cfg():
    return dict(n_vid_in=NLQ_CV, n_txt_in=NLQ_CT, n_embd=NLQ_C, n_head=NLQ_H, n_embd_ks=3, max_len=NLQ_T, arch=(2, 1, 1, 1, 2),
                mha_win_size=[NLQ_WIN, NLQ_WIN, -1, -1], scale_factor=2, with_ln=True, path_pdrop=0.1, use_abs_pe=True)


def nlq_backbone_inputs():
    g = torch.Generator().manual_seed(321)
    vid = torch.randn(2, NLQ_CV, NLQ_T, generator=g)
    vmask = (torch.arange(NLQ_T)[None, :] < torch.tensor([NLQ_T, NLQ_T - 29])[:, None]).unsqueeze(1)
    txt = torch.randn(2, NLQ_CT, NLQ_L, generator=g)
    tmask = (torch.arange(NLQ_L)[None, :] < torch.tensor([NLQ_L - 4, NLQ_L])[:, None]).unsqueeze(1)
    return vid * vmask, vmask, txt * tmask, tmask


# ------------------------------------------------------------------------------------------ NLQ meta-architecture case
NLQ_M_T, NLQ_M_LEVELS = 96, 4


def nlq_model_cfg():
    """kwargs of NLQ's LocPointTransformer (NLQ/libs/modeling/meta_archs.py:345-380) for a small model: the backbone of
    nlq_backbone_cfg() (arch (2,1,1,1,2): 4 pyramid levels, windows 9,9,-1,-1), one query class, label smoothing 0.1"""
    return dict(
        backbone_type='convTransformer', fpn_type='identity', backbone_arch=(2, 1, 1, 1, 2), scale_factor=2,
        input_vid_dim=NLQ_CV, input_txt_dim=NLQ_CT, max_seq_len=NLQ_M_T, max_buffer_len_factor=4.0, n_head=NLQ_H,
        n_mha_win_size=[NLQ_WIN, NLQ_WIN, -1, -1], embd_kernel_size=3, embd_dim=NLQ_C, embd_with_ln=True, fpn_dim=NLQ_C,
        fpn_with_ln=True, fpn_start_level=0, head_dim=NLQ_C, regression_range=[(0, 4), (2, 8), (4, 16), (8, 10000)],
        head_num_layers=3, head_kernel_size=3, head_with_ln=True, use_abs_pe=True, use_rel_pe=False, num_classes=1,
        train_cfg=dict(center_sample='radius', center_sample_radius=1.5, loss_weight=1.0, cls_prior_prob=0.01,
                       init_loss_norm=200, clip_grad_l2norm=1.0, head_empty_cls=[], dropout=0.0, droppath=0.0,
                       label_smoothing=0.1),
        test_cfg=dict(pre_nms_thresh=0.001, pre_nms_topk=2000, iou_threshold=0.1, min_score=0.001, max_seg_num=5,
                      nms_method='soft', nms_sigma=0.75, duration_thresh=0.001, multiclass_nms=True, ext_score_file=None,
                      voting_thresh=0.9),
        cl_cfg=dict(name='mem', memory_size=10, adv_lambda=0, type_sampling='icarl', prompt_pool=False, pool_size=10,
                    topk=4, length=20, embed_dim=NLQ_CT, narration_ssl=False, narration_dim=NLQ_CT, ssl_factor=0.03,
                    use_adapter=False, adapt_blocks=[], reg_lambda=0))


def nlq_model_batch():
    """two query / clip pairs: video features [C, t], query tokens [Ct, L], segments on the feature grid, one-hot labels"""
    g = torch.Generator().manual_seed(4321)
    out = []
    for i, (t, L, segs) in enumerate(((NLQ_M_T, 9, [[10.0, 31.5]]), (NLQ_M_T - 23, 6, [[3.0, 9.25], [40.0, 66.0]]))):
        out.append({'video_id': 'q%d' % i, 'feats': torch.randn(NLQ_CV, t, generator=g),
                    'query_feats': torch.randn(NLQ_CT, L, generator=g), 'segments': torch.tensor(segs),
                    'one_hot_labels': torch.ones(len(segs), 1), 'fps': 30.0, 'duration': 60.0 + i,
                    'feat_stride': 16.043, 'feat_num_frames': 16.043})
    return out


# ------------------------------------------------------------------------------------------------------------------
# iCaRL case: the reference's `classify` (meta_archs.py:1061-1131) only runs on a 10-level pyramid (fpn_levels = 10 is
# hard-coded there), and matches class-distance rows with class logits by flat index, i.e. needs one exemplar class per
# output class.
IC_T, IC_D, IC_CIN, IC_CTXT, IC_NCLS = 1024, 32, 40, 24, 4


def icarl_overrides():
    o = overrides(D=IC_D, T=IC_T, Cin=IC_CIN, Ctxt=IC_CTXT, H=4, use_xl=False, droppath=0.0, cl=dict(name='icarl'))
    o['dataset']['num_classes'] = IC_NCLS
    o['model']['backbone_arch'] = (2, 2, 9)
    o['model']['regression_range'] = [(0, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 512),
                                      (512, 1024), (1024, 10000)]
    o['test_cfg'] = dict(o['test_cfg'], pre_nms_topk=300)
    return o


def icarl_clip(idx, labels=(0, 1)):
    g = torch.Generator().manual_seed(9000 + idx)
    t = IC_T - 31 * (idx % 3)
    return {'video_id': 'ic%d' % idx, 'feats': torch.randn(IC_CIN, t, generator=g),
            'segments': torch.tensor([[20.0 + idx % 7, 90.0], [200.5, 350.25]]), 'labels': torch.tensor(list(labels)),
            'fps': 30.0, 'duration': 300.0, 'feat_stride': 16, 'feat_num_frames': 16,
            'segmentation_labels': torch.zeros(t, IC_NCLS), 'prompt_feature': torch.randn(IC_CTXT, 7 + idx % 3, generator=g)}


def icarl_memory():
    """{class: [exemplar clips]}: two per class, insertion order = class order"""
    return {c: [icarl_clip(10 * c + k, (c, (c + 1) % IC_NCLS)) for k in range(2)] for c in range(IC_NCLS)}


# ------------------------------------------------------------------------------------------------------------------
# NLQ episode case (BASELINE configs[3] scaled down): 3 query-template tasks of the NLQ driver (NLQ/train_cl.py:177-342) on
# the small model of nlq_model_cfg(): per task a fresh optimizer with the head / backbone learning-rate groups
# (NLQ/libs/utils/train_utils.py:63-240, backbone_lr_weight != 1), warm-up + cosine schedule per iteration, replay memory
# (memory_size // 13 queries per template), validation records in the evaluator's input format (:735-746).
NLQ_EP_TASKS, NLQ_EP_PER_TASK, NLQ_EP_BATCH = 3, 4, 2
NLQ_EP_MEMORY = 26                      # -> m = 26 // 13 = 2 queries kept per template


def nlq_episode_opt(backbone_lr_weight=0.5):
    return dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-3, backbone_lr_weight=backbone_lr_weight,
                coef_lr=1, epochs=1, warmup=True, warmup_epochs=1, schedule_type="cosine", schedule_steps=[], schedule_gamma=0.1)


def nlq_episode_query(task, k):
    """query k of template `task`: clip features [Cv, t], query tokens [Ct, L], one or two moments, evaluator ids"""
    idx = 10 * task + k
    g = torch.Generator().manual_seed(7000 + idx)
    t = NLQ_M_T - 7 * (idx % 4)
    L = 5 + idx % 5
    s0 = 4.0 + 3 * (idx % 6)
    segs = [[s0, s0 + 6.5 + idx % 3]] + ([[50.0, 70.25 + idx % 4]] if idx % 2 else [])
    return {'video_id': 'clip%02d' % idx, 'query_id': 'ann%02d_%d' % (idx, k), 'feats': torch.randn(NLQ_CV, t, generator=g),
            'query_feats': torch.randn(NLQ_CT, L, generator=g), 'segments': torch.tensor(segs),
            'one_hot_labels': torch.ones(len(segs), 1), 'fps': 30.0, 'duration': 60.0 + idx, 'feat_stride': 16.043,
            'feat_num_frames': 16.043}


def nlq_episode_data(task):
    """{template name: [query dicts]} of task `task` (the value of the reference's data['train'][template])"""
    return {'template_%d' % task: [nlq_episode_query(task, k) for k in range(NLQ_EP_PER_TASK)]}


def nlq_episode_metric(results):
    """stand-in for ReferringRecall (the evaluator is outside the hot path): mean top-1 confidence of the records"""
    return float(sum(r['predicted_times'][0][2] for r in results) / max(len(results), 1))


# ------------------------------------------------------------------------------------------------------------------
# Evaluator-format case: two validation tasks of two clips each on the "xl" golden model
def eval_clips(task):
    okw, L = CASES['xl']
    o = overrides(**okw)
    vl = video_list(o['dataset']['max_seq_len'], o['dataset']['input_dim'], o['model']['n_txt_in'], L, seed=50 + task)
    for i, v in enumerate(vl):
        v['video_id'] = 'val%d_%d' % (task, i)
    return vl


def eval_fake_map(results):
    """stand-in for ANETdetection.evaluate (outside the hot path): a number determined by the records"""
    import numpy as np
    return float(np.mean(results['score'])) if len(results['score']) else 0.0


def eval_fake_recall(json_obj):
    """stand-in for evaluation_retrieval: a [5 tIoU, 2 ranks] table determined by the JSON the path wrote"""
    import numpy as np
    rows = [r for v in json_obj['results'].values() for r in v]
    s = float(np.mean([r['score'] for r in rows])) if rows else 0.0
    d = float(np.mean([r['segment'][1] - r['segment'][0] for r in rows])) if rows else 0.0
    return np.array([[s * (i + 1) / 5.0, d / (10.0 * (i + 1))] for i in range(5)])
