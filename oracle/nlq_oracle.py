"""TEST INFRASTRUCTURE ONLY -- never imported by vilco_amd/ (the product path).

CPU restatement of the NLQ model variant's backbone path (reference: NLQ/libs/modeling/blocks.py LocalMaskedMHCA
:417-755, TransformerBlock :757-875; backbones.py ConvTransformerBackbone :409-615), functional style like
oracle/mq_oracle.py (flat parameter dict + key prefix, channel-first activations, bool masks [B,1,T]).

The sliding-window attention is restated as what it computes -- a banded softmax -- not as the Longformer chunking
the reference uses to evaluate it.  Parity pinned: tests/test_nlq.py compares this file with tests/golden/nlq_blocks.pt,
recorded from the imported reference blocks (tests/golden/make_golden_nlq.py)."""
import math

import torch
import torch.nn.functional as F

from . import mq_oracle as M


def local_mhca(p, pre, x, mask, n_head, stride, window_size):
    """LocalMaskedMHCA.forward (blocks.py:692-755): keys j with |i - j| <= window_size // 2 inside the sequence
    (:541-553 masks the positions beyond the ends), -1e4 added to masked keys (:722-735), masked queries zeroed (:740)."""
    C = x.shape[1]
    w = window_size // 2
    q, qx_mask = M.masked_conv1d(x, mask, p[pre + 'query_conv.conv.weight'], None, stride, C)
    q = M.ln(p, pre + 'query_norm.', q)
    k, kv_mask = M.masked_conv1d(x, mask, p[pre + 'key_conv.conv.weight'], None, stride, C)
    k = M.ln(p, pre + 'key_norm.', k)
    v, _ = M.masked_conv1d(x, mask, p[pre + 'value_conv.conv.weight'], None, stride, C)
    v = M.ln(p, pre + 'value_norm.', v)
    q, k, v = M.conv1x1(p, pre + 'query.', q), M.conv1x1(p, pre + 'key.', k), M.conv1x1(p, pre + 'value.', v)
    B, _, T = q.shape
    scale = 1.0 / math.sqrt(C // n_head)
    qh, kh, vh = M._heads(q, n_head), M._heads(k, n_head), M._heads(v, n_head)
    att = (qh * scale) @ kh.transpose(-2, -1)
    idx = torch.arange(T)
    band = (idx[:, None] - idx[None, :]).abs() <= w
    km = kv_mask[:, 0, :]
    att = att.masked_fill(torch.logical_not(band)[None, None], float('-inf'))
    att = att + (-1e4) * torch.logical_not(km)[:, None, None, :].to(att.dtype)
    att = F.softmax(att, dim=-1)
    att = att.masked_fill(torch.logical_not(km)[:, None, :, None], 0.0)
    out = (att @ vh).transpose(2, 3).contiguous().view(B, C, -1)
    return M.conv1x1(p, pre + 'proj.', out) * qx_mask.to(out.dtype), qx_mask


def transformer_block(p, pre, x, mask, n_head, stride, window_size=-1, cross_y=None, cross_mask=None):
    """NLQ TransformerBlock.forward (blocks.py:851-875), deterministic (eval-mode dropout / drop path)."""
    def dp(name, v):
        key = pre + name + '.scale'
        return p[key] * v if key in p else v
    h = M.ln(p, pre + 'ln1.', x)
    if window_size > 1:
        out, out_mask = local_mhca(p, pre + 'attn.', h, mask, n_head, stride, window_size)
    else:
        out, out_mask = M.masked_mhca(p, pre + 'attn.', h, mask, n_head, stride)
    mf = out_mask.to(out.dtype)
    skip = F.max_pool1d(x, stride + 1, stride=stride, padding=(stride + 1) // 2) if stride > 1 else x
    out = skip * mf + dp('drop_path_attn', out)
    if cross_y is not None and (pre + 'cross_attn.query.weight') in p:
        c = M.masked_mha(p, pre + 'cross_attn.', M.ln(p, pre + 'ln3.', out), mf, M.ln(p, pre + 'ln3.', cross_y), cross_mask, n_head)
        out = out * mf + dp('drop_path_attn', c)
    m = F.conv1d(F.gelu(M.conv1x1(p, pre + 'mlp.0.', M.ln(p, pre + 'ln2.', out))), p[pre + 'mlp.3.weight'], p[pre + 'mlp.3.bias'])
    return out + dp('drop_path_mlp', m * mf), out_mask


def backbone(p, cfg, vid, vid_mask, txt, txt_mask, training=True, pre=''):
    """ConvTransformerBackbone.forward (backbones.py:551-615).  cfg: the constructor kwargs."""
    arch, n_head, wins = cfg['arch'], cfg['n_head'], cfg['mha_win_size']
    T = vid.shape[-1]
    for i in range(arch[0]):
        vid, vid_mask = M.masked_conv1d(vid, vid_mask, p[pre + 'vid_embd.%d.conv.weight' % i], p.get(pre + 'vid_embd.%d.conv.bias' % i))
        if cfg['with_ln']:
            vid = M.ln(p, pre + 'vid_embd_norm.%d.' % i, vid)
        vid = torch.relu(vid)
    if cfg['use_abs_pe']:
        pe = M.sinusoid_pe(cfg['max_len'], cfg['n_embd'], vid.dtype)
        if (not training) and T >= cfg['max_len']:
            pe = F.interpolate(pe, T, mode='linear', align_corners=False)
        vid = vid + pe[:, :, :T] * vid_mask.to(vid.dtype)
    for i in range(arch[0]):
        txt, txt_mask = M.masked_conv1d(txt, txt_mask, p[pre + 'txt_embd.%d.conv.weight' % i], p.get(pre + 'txt_embd.%d.conv.bias' % i))
        if cfg['with_ln']:
            txt = M.ln(p, pre + 'txt_embd_norm.%d.' % i, txt)
        txt = torch.relu(txt)
    for i in range(arch[1]):
        txt, txt_mask = transformer_block(p, pre + 'txt_stem.%d.' % i, txt, txt_mask, n_head, 1)
    qm = txt_mask.squeeze(1)
    for i in range(arch[2]):
        vid, vid_mask = transformer_block(p, pre + 'vid_stem.%d.' % i, vid, vid_mask, n_head, 1, wins[0], txt, qm)
    feats, masks = [vid], [vid_mask]
    for i in range(arch[3] + arch[4]):
        idx = i if i < arch[3] else i - arch[3]                      # the reference restarts the window index (:520-545)
        cross = i < arch[3]
        vid, vid_mask = transformer_block(p, pre + 'branch.%d.' % i, vid, vid_mask, n_head, cfg['scale_factor'], wins[1 + idx],
                                          txt if cross else None, qm if cross else None)
        feats.append(vid)
        masks.append(vid_mask)
    return feats, masks


# ------------------------------------------------------------------------------------------ meta-architecture
# NLQ/libs/modeling/meta_archs.py: heads :182-337 (the MQ heads), label_points_single_video :981-1068 (no gaussian point
# weights, labels arrive one-hot), losses :1094-1198 (focal with label smoothing + DIoU, no action-localisation term),
# inference_single_video :1253-1338, postprocessing :1341-1382.
def backbone_cfg(cfg):
    """the kwargs PtTransformer.__init__ hands to make_backbone (:440-460)"""
    n_levels = cfg['backbone_arch'][-2] + cfg['backbone_arch'][-1] + 1
    win = cfg['n_mha_win_size']
    return dict(n_vid_in=cfg['input_vid_dim'], n_txt_in=cfg['input_txt_dim'], n_embd=cfg['embd_dim'], n_head=cfg['n_head'],
                n_embd_ks=cfg['embd_kernel_size'], max_len=cfg['max_seq_len'], arch=cfg['backbone_arch'],
                mha_win_size=[win] * n_levels if isinstance(win, int) else win, scale_factor=cfg['scale_factor'],
                with_ln=cfg['embd_with_ln'], use_abs_pe=cfg['use_abs_pe'])


def level_strides(cfg):
    n_levels = cfg['backbone_arch'][-2] + cfg['backbone_arch'][-1] + 1
    return [cfg['scale_factor'] ** i for i in range(cfg['fpn_start_level'], n_levels)]


def points(cfg, level_lens, dtype=torch.float32):
    """PointGenerator buffers (loc_generators.py): [T_l, 4] = (t, reg_lo, reg_hi, stride)"""
    out = []
    max_len = cfg['max_seq_len'] * cfg['max_buffer_len_factor']
    for n, s, rr in zip(level_lens, level_strides(cfg), cfg['regression_range']):
        t = torch.arange(0, max_len, s)[:, None].float()
        k = t.shape[0]
        pts = torch.cat((t, torch.as_tensor(rr, dtype=torch.float)[None].repeat(k, 1), torch.full((k, 1), float(s))), dim=1)
        out.append(pts[:n].to(dtype))
    return out


def batch_inputs(cfg, video_list, training, dtype=torch.float32):
    """preprocessing :919-957 + query_preprocessing :879-916"""
    feats = [x['feats'].to(dtype) for x in video_list]
    lens = torch.as_tensor([f.shape[-1] for f in feats])
    max_len = int(lens.max())
    if training:
        assert max_len <= cfg['max_seq_len']
        max_len = cfg['max_seq_len']
    else:
        assert len(video_list) == 1
        if max_len <= cfg['max_seq_len']:
            max_len = cfg['max_seq_len']
        else:
            stride = 1
            for s, w in zip(level_strides(cfg), backbone_cfg(cfg)['mha_win_size']):
                stride = max(stride, s * (w // 2) * 2 if w > 1 else s)
            max_len = (max_len + stride - 1) // stride * stride
    vid = feats[0].new_zeros(len(feats), feats[0].shape[0], max_len)
    for f, dst in zip(feats, vid):
        dst[..., :f.shape[-1]].copy_(f)
    vmask = (torch.arange(max_len)[None, :] < lens[:, None]).unsqueeze(1)
    q = [x['query_feats'].to(dtype) for x in video_list]
    qlens = torch.as_tensor([f.shape[-1] for f in q])
    txt = q[0].new_zeros(len(q), q[0].shape[0], int(qlens.max()))
    for f, dst in zip(q, txt):
        dst[..., :f.shape[-1]].copy_(f)
    tmask = (torch.arange(int(qlens.max()))[None, :] < qlens[:, None]).unsqueeze(1)
    return vid, vmask, txt, tmask


def forward_network(p, cfg, video_list, training=True):
    vid, vmask, txt, tmask = batch_inputs(cfg, video_list, training, next(iter(p.values())).dtype)
    feats, masks = backbone(p, backbone_cfg(cfg), vid, vmask, txt, tmask, training, pre='backbone.')
    feats, masks = M.neck(p, cfg, feats, masks)
    cls, reg = M.heads(p, cfg, feats, masks)
    return masks, cls, reg


def label_points_single(cfg, pts, seg, one_hot):
    """label_points_single_video :981-1068"""
    tc = cfg['train_cfg']
    n = pts.shape[0]
    if seg.shape[0] == 0:
        return seg.new_zeros((n, cfg['num_classes'])), seg.new_zeros((n, 2))
    t, stride = pts[:, 0, None], pts[:, 3, None]
    lens = (seg[:, 1] - seg[:, 0])[None, :].repeat(n, 1)
    left, right = t - seg[None, :, 0], seg[None, :, 1] - t
    reg = torch.stack((left, right), dim=-1)
    if tc['center_sample'] == 'radius':
        ctr = 0.5 * (seg[None, :, 0] + seg[None, :, 1])
        lo = t - torch.maximum(ctr - stride * tc['center_sample_radius'], seg[None, :, 0])
        hi = torch.minimum(ctr + stride * tc['center_sample_radius'], seg[None, :, 1]) - t
        inside = torch.stack((lo, hi), -1).min(-1)[0] > 0
    else:
        inside = reg.min(-1)[0] > 0
    far = reg.max(-1)[0]
    in_range = torch.logical_and(far >= pts[:, 1, None], far <= pts[:, 2, None])
    lens = lens.masked_fill(inside == 0, float('inf')).masked_fill(in_range == 0, float('inf'))
    min_len, idx = lens.min(dim=1)
    sel = torch.logical_and(lens <= (min_len[:, None] + 1e-3), lens < float('inf')).to(reg.dtype)
    cls_t = (sel @ one_hot.to(reg.dtype)).clamp(min=0.0, max=1.0)
    return cls_t, reg[torch.arange(n), idx] / stride


def losses(cfg, masks, cls_logits, offsets, segments, one_hots, loss_normalizer):
    """PtTransformer.losses :1094-1198 (no CL distillation terms) -> (dict, new loss_normalizer)"""
    tc = cfg['train_cfg']
    dt = cls_logits[0].dtype
    pts = torch.cat(points(cfg, [c.shape[1] for c in cls_logits], dt), dim=0)
    lab = [label_points_single(cfg, pts, s.to(dt), o) for s, o in zip(segments, one_hots)]
    gt_cls, gt_off = torch.stack([x[0] for x in lab]), torch.stack([x[1] for x in lab])
    valid = torch.cat([m.squeeze(1) for m in masks], dim=1)
    pos = torch.logical_and(gt_cls.sum(-1) > 0, valid)
    num_pos = int(pos.sum())
    loss_normalizer = 0.9 * loss_normalizer + 0.1 * max(num_pos, 1)
    ls = tc['label_smoothing']
    target = gt_cls[valid] * (1 - ls) + ls / (gt_cls.shape[-1] + 1)
    cls_loss = M.sigmoid_focal(torch.cat(cls_logits, dim=1)[valid], target).sum() / loss_normalizer
    pred = torch.cat(offsets, dim=1)[pos]
    reg_loss = 0 * pred.sum() if num_pos == 0 else M.diou_1d(pred, gt_off[pos]).sum() / loss_normalizer
    lw = tc['loss_weight'] if tc['loss_weight'] > 0 else cls_loss.detach() / max(float(reg_loss), 0.01)
    return {'cls_loss': cls_loss, 'reg_loss': reg_loss, 'final_loss': cls_loss + reg_loss * lw}, loss_normalizer


def forward_losses(p, cfg, video_list, loss_normalizer=None):
    masks, cls, reg = forward_network(p, cfg, video_list, True)
    ln0 = cfg['train_cfg']['init_loss_norm'] if loss_normalizer is None else loss_normalizer
    return losses(cfg, masks, cls, reg, [x['segments'] for x in video_list], [x['one_hot_labels'] for x in video_list], ln0)


def decode(cfg, masks, cls_logits, offsets):
    """inference_single_video :1253-1338 of the single clip of an eval batch -> (segments, scores, labels) before NMS"""
    pts = points(cfg, [c.shape[1] for c in cls_logits], cls_logits[0].dtype)
    return M.decode_single_video(cfg, pts, [m[0].squeeze(0) for m in masks], [c[0] for c in cls_logits], [o[0] for o in offsets])
