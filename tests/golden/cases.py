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
