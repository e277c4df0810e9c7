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
