"""TEST INFRASTRUCTURE ONLY -- never imported by vilco_amd/ (the product path).

CPU restatement, in plain PyTorch tensor expressions, of the reference's MQ training / inference
path: MQ/libs/modeling/{blocks,backbones,modeling_xlnet_x,necks,loc_generators,meta_archs,losses}.py.
It is the checker the HIP path is compared with on the GPU box (where /root/reference does not exist)
and the `cpu_baseline` ("port") that bench.py times.

Functional style: every function takes the flat parameter dict `p` (a reference-format state_dict,
any float dtype -- float64 gives a tighter checker) plus a key prefix; activations are channel-first
[B, C, T] with bool masks [B, 1, T], exactly like the reference.  Each function cites the reference
lines it follows.

Parity pinned: tests/test_oracle_model.py checks this file (a) against the imported reference itself
when /root/reference is present and (b) against the committed golden vectors tests/golden/model_*.pt
that tests/golden/make_golden.py generated from the imported reference (losses, every parameter
gradient, inference outputs; <= 1e-5 relative in fp32).
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------ dropout contexts
# The reference trains with nn.Dropout / stochastic depth on (mq_vilco.yaml: dropout 0.1, droppath 0.1; XLNet 0.1).
# DROP = None restates eval-mode arithmetic.  A context object supplies the MASK FACTORS (0 or 1/(1-p)) per site:
#   DropReplay  -- the factors the HIP path drew (ops.dropout_log), consumed per site in call order (parity tests);
#   DropRandom  -- fresh Bernoulli masks, i.e. the arithmetic the reference does in train mode (cpu_baseline timing).
# Sites: 'proj_drop' (blocks.py:264,405), 'mlp_drop' (:533-540, two per block), 'droppath' (per-sample factors [B],
# :628-641), and XLNet's seven sites (see xlnet_layer).  Factors arrive token-major ([B,T,C]) like the HIP tensors.
DROP = None


class DropReplay:
    def __init__(self, log, mask_fn):
        """log: ops.dropout_log entries (site, p, seed, shape) or (site, tensor); mask_fn(p, seed, shape, site) -> factors
        (the attention-probability site has a mask function of its own)"""
        self.q = {}
        for e in log:
            m = e[1] if len(e) == 2 else mask_fn(e[1], e[2], e[3], e[0])
            self.q.setdefault(e[0], []).append(m)
        self.p = {'droppath': 1.0, 'proj_drop': 1.0, 'mlp_drop': 1.0, 'xl': 1.0}     # > 0: every site is live

    def factor(self, site, shape, dtype):
        m = self.q[site].pop(0)
        assert tuple(m.shape) == tuple(shape), (site, tuple(m.shape), tuple(shape))
        return m.to(dtype).cpu()

    def leftover(self):
        return {k: len(v) for k, v in self.q.items() if v}


# ---- sign decisions of the LayerNorm -> ReLU sites handed in from outside (tests/test_fullsize_gpu.py).  A pre-activation within
# rounding of zero takes one side or the other depending on the arithmetic; an implementation that is not bit-identical to this
# restatement can land on the other side of such an element, after which every gradient upstream differs by a discrete amount.
# RELU = ReluReplay(masks) makes this run take the SAME side as the run that produced `masks` (y = x * mask: the output differs
# from relu(x) by the size of the near-zero pre-activation, the derivative is the other run's) and records where the sides
# differed and how far from zero those pre-activations were.  None: plain torch.relu.
RELU = None


class ReluReplay:
    def __init__(self, masks):
        """masks: {site: bool tensor in this oracle's layout [B, C, T]} -- sites without an entry take their own side"""
        self.masks = dict(masks)
        self.events = []          # (site, elements that changed side, max |pre-activation| among them / max |pre-activation|)
        self.seen = set()

    def __call__(self, site, x):
        m = self.masks.get(site)
        if m is None:
            return torch.relu(x)
        assert tuple(m.shape) == tuple(x.shape), (site, tuple(m.shape), tuple(x.shape))
        self.seen.add(site)
        diff = (x.detach() > 0) != m
        n = int(diff.sum())
        if n:
            self.events.append((site, n, float(x.detach().abs()[diff].max() / x.detach().abs().max().clamp_min(1e-30))))
        return x * m.to(x.dtype)


def _relu(site, x):
    return torch.relu(x) if RELU is None else RELU(site, x)


class DropRandom:
    def __init__(self, dropout=0.1, droppath=0.1, xl=0.1, seed=0):
        self.p = {'droppath': droppath, 'proj_drop': dropout, 'mlp_drop': dropout, 'xl': xl}
        self.g = torch.Generator().manual_seed(seed)

    def factor(self, site, shape, dtype):
        p = self.p.get(site, self.p['xl'])
        return torch.bernoulli(torch.full(tuple(shape), 1.0 - p), generator=self.g).to(dtype) / (1.0 - p)


def _live(site):
    return DROP is not None and DROP.p.get(site, DROP.p['xl']) > 0.0


def _drop_cf(site, t):
    """dropout of a channel-first [B,C,T] tensor with token-major factors"""
    if not _live(site):
        return t
    B, Cn, T = t.shape
    return t * DROP.factor(site, (B, T, Cn), t.dtype).permute(0, 2, 1)


def _drop_path(t):
    """per-sample stochastic depth (drop_path, blocks.py:628-641) of [B, ...]"""
    if not _live('droppath'):
        return t
    return t * DROP.factor('droppath', (t.shape[0],), t.dtype).view(-1, *([1] * (t.dim() - 1)))


# ------------------------------------------------------------------------------ operators
def masked_conv1d(x, mask, w, b=None, stride=1, groups=1):
    """MaskedConv1D.forward, blocks.py:106-130: conv, mask[::stride] (nearest), multiply."""
    pad = w.shape[-1] // 2
    y = F.conv1d(x, w, b, stride=stride, padding=pad, groups=groups)
    if stride > 1:
        out_mask = F.interpolate(mask.to(x.dtype), size=x.shape[-1] // stride, mode='nearest')
    else:
        out_mask = mask.to(x.dtype)
    return y * out_mask, out_mask.bool()


def layer_norm_cf(x, w, b, eps=1e-5):
    """LayerNorm over dim 1 of [B,C,T], biased variance, eps inside sqrt (blocks.py:160-175)."""
    r = x - x.mean(dim=1, keepdim=True)
    y = r / torch.sqrt((r ** 2).mean(dim=1, keepdim=True) + eps)
    return y * w + b


def ln(p, pre, x):
    return layer_norm_cf(x, p[pre + 'weight'], p[pre + 'bias'])


def conv1x1(p, pre, x):
    return F.conv1d(x, p[pre + 'weight'], p[pre + 'bias'])


def _heads(x, n_head):
    B, C, T = x.shape
    return x.view(B, n_head, C // n_head, T).transpose(2, 3)


def _attend(q, k, v, key_mask, n_head):
    """softmax((q*scale) k^T masked_fill(-inf)) (v*mask)   (blocks.py:383-400 / 251-265).
    key_mask: bool [B, Tk]."""
    B, C, _ = q.shape
    scale = 1.0 / math.sqrt(C // n_head)
    q, k, v = _heads(q, n_head), _heads(k, n_head), _heads(v, n_head)
    att = (q * scale) @ k.transpose(-2, -1)
    att = att.masked_fill(torch.logical_not(key_mask[:, None, None, :]), float('-inf'))
    att = F.softmax(att, dim=-1)
    out = att @ (v * key_mask[:, None, :, None].to(v.dtype))
    return out.transpose(2, 3).contiguous().view(B, C, -1)


def masked_mhca(p, pre, x, mask, n_head, stride):
    """MaskedMHCA.forward, blocks.py:351-410 (the query conv also uses the kv stride, :313)."""
    C = x.shape[1]
    q, qx_mask = masked_conv1d(x, mask, p[pre + 'query_conv.conv.weight'], None, stride, C)
    q = ln(p, pre + 'query_norm.', q)
    k, kv_mask = masked_conv1d(x, mask, p[pre + 'key_conv.conv.weight'], None, stride, C)
    k = ln(p, pre + 'key_norm.', k)
    v, _ = masked_conv1d(x, mask, p[pre + 'value_conv.conv.weight'], None, stride, C)
    v = ln(p, pre + 'value_norm.', v)
    q, k, v = conv1x1(p, pre + 'query.', q), conv1x1(p, pre + 'key.', k), conv1x1(p, pre + 'value.', v)
    out = _attend(q, k, v, kv_mask[:, 0, :], n_head)
    return _drop_cf('proj_drop', conv1x1(p, pre + 'proj.', out)) * qx_mask.to(out.dtype), qx_mask


def masked_mha(p, pre, x, mask_float, enc, enc_mask, n_head):
    """MaskedMHA.forward as cross-attention, blocks.py:228-269.  enc_mask: [B, L] (long/bool)."""
    q = conv1x1(p, pre + 'query.', x)
    k = conv1x1(p, pre + 'key.', enc)
    v = conv1x1(p, pre + 'value.', enc)
    out = _attend(q, k, v, enc_mask.bool(), n_head)
    return _drop_cf('proj_drop', conv1x1(p, pre + 'proj.', out)) * mask_float


def channel_attention(p, pre, x, n_head):
    """ChannelAttention.forward on [B,T,C], blocks.py:423-436."""
    B, T, C = x.shape
    qkv = F.linear(x, p[pre + 'qkv.weight'], p.get(pre + 'qkv.bias'))
    qkv = qkv.reshape(B, T, 3, n_head, C // n_head).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = ((k * (C // n_head) ** -0.5).transpose(-1, -2) @ v).softmax(dim=-1)
    y = (att @ q.transpose(-1, -2)).transpose(-1, -2)
    y = y.transpose(1, 2).reshape(B, T, C)
    return F.linear(y, p[pre + 'proj.weight'], p[pre + 'proj.bias'])


def channel_block(p, pre, x, n_head):
    """ChannelBlock.forward, blocks.py:459-466: norm1 is never applied, no mask."""
    x = x.permute(0, 2, 1)
    x = x + _drop_path(channel_attention(p, pre + 'attn.', x, n_head))
    h = F.layer_norm(x, (x.shape[-1],), p[pre + 'norm2.weight'], p[pre + 'norm2.bias'], 1e-5)
    h = F.linear(F.gelu(F.linear(h, p[pre + 'mlp.0.weight'], p[pre + 'mlp.0.bias'])),
                 p[pre + 'mlp.2.weight'], p[pre + 'mlp.2.bias'])
    return (x + _drop_path(h)).permute(0, 2, 1)


def adapter(p, pre, x):
    """Adapter.layer on the TIME axis of [B,C,T] (meta_archs.py:123-148, 'parallel' mode)."""
    h = F.gelu(F.linear(x, p[pre + 'layer.0.weight'], p[pre + 'layer.0.bias']))
    return F.linear(h, p[pre + 'layer.2.weight'], p[pre + 'layer.2.bias'])


def transformer_block(p, pre, x, mask, n_head, stride, t_c_alpha, cross_y=None, cross_mask=None,
                      adapter_pre=None):
    """TransformerBlock.forward, blocks.py:561-593 (dropout / stochastic depth through the DROP context)."""
    def dp(name, v):          # AffineDropPath: scale then drop_path (blocks.py:669-670); Identity when droppath == 0
        key = pre + name + '.scale'
        return _drop_path(p[key] * v) if key in p else v

    h = ln(p, pre + 'ln1.', x)
    out, out_mask = masked_mhca(p, pre + 'attn.', h, mask, n_head, stride)
    if adapter_pre is not None:
        out = out + adapter(p, adapter_pre, h)
    mf = out_mask.to(out.dtype)
    skip = F.max_pool1d(x, stride + 1, stride=stride, padding=(stride + 1) // 2) if stride > 1 else x
    out = skip * mf + dp('drop_path_attn', out)
    if cross_y is not None and (pre + 'cross_attn.query.weight') in p:
        c = masked_mha(p, pre + 'cross_attn.', ln(p, pre + 'ln3.', out), mf, ln(p, pre + 'ln3.', cross_y),
                       cross_mask, n_head)
        out = out * mf + dp('drop_path_attn', c)
    m = _drop_cf('mlp_drop', F.gelu(conv1x1(p, pre + 'mlp.0.', ln(p, pre + 'ln2.', out))))
    m = _drop_cf('mlp_drop', F.conv1d(m, p[pre + 'mlp.3.weight'], p[pre + 'mlp.3.bias']))
    out = out + dp('drop_path_mlp', m * mf)
    if stride == 1:
        out = t_c_alpha * out + (1 - t_c_alpha) * channel_block(p, pre + 'channel_attn.', h, n_head)
    return out, out_mask


# ------------------------------------------------------------------------------ XLNet layer
def xlnet_layer(p, pre, x, mask, n_head, drop=None):
    """XLNetModel.forward (bi, no mems / segments / target mapping) for one layer:
    modeling_xlnet_x.py:1075-1308, rel_attn :437-461 -> rel_attn_core :270-320, post_attention :322-332,
    XLNetFeedForward :482-490.  x [B,T,D], mask [B,T] (1 = valid) -> [B,T,D].
    drop: None (eval / p = 0) or a dict of dropout MASK FACTORS (0 or 1/(1-p)) for the reference's seven nn.Dropout
    sites, in batch-major layouts: 'xl_input' [B,T,D] (:1201), 'xl_pos_emb' [B,2T,D] (:1228), 'attn_prob' [B,H,T,T]
    (:308), 'xl_attn_out' [B,T,D] (:327), 'xl_ff_inner' [B,T,d_inner] (:486), 'xl_ff_out' [B,T,D] (:488),
    'xl_output' [B,T,D] (:1280)."""
    B, T, D = x.shape
    if drop is None and _live('xl'):
        d_inner = p[pre + 'ff.layer_1.weight'].shape[0]
        shapes = {'xl_input': (B, T, D), 'xl_pos_emb': (B, 2 * T, D), 'attn_prob': (B, n_head, T, T),
                  'xl_attn_out': (B, T, D), 'xl_ff_inner': (B, T, d_inner), 'xl_ff_out': (B, T, D),
                  'xl_output': (B, T, D)}
        drop = {k: DROP.factor(k, sh, x.dtype) for k, sh in shapes.items()}      # forward order of the seven sites
    dm = (lambda k, t: t) if drop is None else (lambda k, t: t * drop[k].to(t.dtype))
    x = dm('xl_input', x)
    h = x.transpose(0, 1)                                           # [T,B,D]
    att = mask.transpose(0, 1).to(x.dtype)                          # [T,B]
    data_mask = (1.0 - att)[None]                                   # [1,T,B]
    attn_mask = (data_mask[:, :, :, None] > 0).to(x.dtype)
    non_tgt = ((attn_mask + (-torch.eye(T, dtype=x.dtype))[:, :, None, None]) > 0).to(x.dtype)   # [T,T,B,1]

    freq_seq = torch.arange(0, D, 2.0, dtype=torch.float)
    inv_freq = 1 / torch.pow(10000, (freq_seq / D))
    pos_seq = torch.arange(T, -T, -1.0, dtype=torch.float)
    sinus = torch.einsum("i,d->id", pos_seq, inv_freq)
    pos_emb = torch.cat([torch.sin(sinus), torch.cos(sinus)], dim=-1)[:, None, :].expand(-1, B, -1).to(x.dtype)
    if drop is not None:
        pos_emb = pos_emb * drop['xl_pos_emb'].to(x.dtype).transpose(0, 1)

    a = pre + 'rel_attn.'
    q = torch.einsum("ibh,hnd->ibnd", h, p[a + 'q'])
    k = torch.einsum("ibh,hnd->ibnd", h, p[a + 'k'])
    v = torch.einsum("ibh,hnd->ibnd", h, p[a + 'v'])
    kr = torch.einsum("ibh,hnd->ibnd", pos_emb, p[a + 'r'])
    ac = torch.einsum("ibnd,jbnd->bnij", q + p[a + 'r_w_bias'], k)
    bd = torch.einsum("ibnd,jbnd->bnij", q + p[a + 'r_r_bias'], kr)
    s = bd.shape                                                    # rel_shift_bnij, :256-268
    bd = bd.reshape(s[0], s[1], s[3], s[2])[:, :, 1:, :].reshape(s[0], s[1], s[2], s[3] - 1)
    bd = torch.index_select(bd, 3, torch.arange(T, dtype=torch.long))
    d_head = p[a + 'q'].shape[-1]
    score = (ac + bd) * (1 / (d_head ** 0.5)) - 1e30 * torch.einsum("ijbn->bnij", non_tgt)
    prob = dm('attn_prob', F.softmax(score, dim=3))
    vec = torch.einsum("bnij,jbnd->ibnd", prob, v)
    out = torch.einsum("ibnd,hnd->ibh", vec, p[a + 'o'])
    out = dm('xl_attn_out', out.transpose(0, 1)).transpose(0, 1) + h
    out = F.layer_norm(out, (D,), p[a + 'layer_norm.weight'], p[a + 'layer_norm.bias'], 1e-12)
    f = pre + 'ff.'
    y = F.gelu(F.linear(out, p[f + 'layer_1.weight'], p[f + 'layer_1.bias']))
    y = dm('xl_ff_inner', y.transpose(0, 1)).transpose(0, 1)
    y = F.linear(y, p[f + 'layer_2.weight'], p[f + 'layer_2.bias'])
    y = dm('xl_ff_out', y.transpose(0, 1)).transpose(0, 1)
    y = F.layer_norm(y + out, (D,), p[f + 'layer_norm.weight'], p[f + 'layer_norm.bias'], 1e-12)
    return dm('xl_output', y.permute(1, 0, 2).contiguous())


# ------------------------------------------------------------------------------ model
def sinusoid_pe(n_position, d_hid, dtype):
    """get_sinusoid_encoding(max_len, D) / sqrt(D)  (blocks.py:179-190, backbones.py:62) -> [1,D,T]."""
    pos = torch.arange(n_position, dtype=torch.float64)[:, None]
    j = torch.arange(d_hid)
    ang = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2 * torch.div(j, 2, rounding_mode='floor') / d_hid)
    tab = torch.where(j % 2 == 0, torch.sin(ang), torch.cos(ang)).float()
    return (tab.unsqueeze(0).transpose(1, 2) / (d_hid ** 0.5)).to(dtype)


def backbone(p, cfg, x, mask, text=None, text_mask=None, training=True, adapter_blocks=(), pets_prefix='pets.'):
    """ConvTransformerBackbone.forward, backbones.py:181-289."""
    m = cfg
    pre = 'backbone.'
    n_head, arch, alpha = m['n_head'], m['backbone_arch'], m['train_cfg']['t_c_alpha']
    T = x.shape[-1]
    for i in range(arch[0]):
        x, mask = masked_conv1d(x, mask, p[pre + 'embd.%d.conv.weight' % i], p.get(pre + 'embd.%d.conv.bias' % i))
        if m['embd_with_ln']:
            x = ln(p, pre + 'embd_norm.%d.' % i, x)
        x = _relu(pre + 'embd_norm.%d' % i, x)
    if m['use_abs_pe']:
        pe = sinusoid_pe(m['max_seq_len'], m['embd_dim'], x.dtype)
        if (not training) and T >= m['max_seq_len']:
            pe = F.interpolate(pe, T, mode='linear', align_corners=False)
        x = x + pe[:, :, :T] * mask.to(x.dtype)
    q = q_mask = None
    if m['use_cross_modal'] and text is not None:
        q, qm = text, text_mask
        for i in range(arch[0]):
            q, qm = masked_conv1d(q, qm, p[pre + 'txt_embd.%d.conv.weight' % i], p.get(pre + 'txt_embd.%d.conv.bias' % i))
            if m['embd_with_ln']:
                q = ln(p, pre + 'txt_embd_norm.%d.' % i, q)
            q = _relu(pre + 'txt_embd_norm.%d' % i, q)
        for i in range(arch[1]):
            q, qm = transformer_block(p, pre + 'txt_stem.%d.' % i, q, qm, n_head, 1, 0.8)
        q_mask = qm.squeeze(1).long()
    for i in range(arch[1]):
        x, mask = transformer_block(p, pre + 'stem.%d.' % i, x, mask, n_head, 1, alpha)
    feats, masks = [x], [mask]
    for i in range(arch[2]):
        if i == 0:
            if m['use_xl']:
                x = xlnet_layer(p, pre + 'xlnet.layer.0.', x.permute(0, 2, 1), mask.squeeze(1).long(),
                                p[pre + 'xlnet.layer.0.rel_attn.q'].shape[1]).permute(0, 2, 1)
            else:
                x, mask = transformer_block(p, pre + 'stem.0.', x, mask, n_head, 1, alpha)
        ad = None
        if i in adapter_blocks:
            ad = pets_prefix + '%d.' % list(adapter_blocks).index(i)
        if i in (1, 2):
            x, mask = transformer_block(p, pre + 'branch.%d.' % i, x, mask, n_head, m['scale_factor'], alpha,
                                        adapter_pre=ad)
        else:
            x, mask = transformer_block(p, pre + 'branch.%d.' % i, x, mask, n_head, m['scale_factor'], alpha, q,
                                        q_mask, adapter_pre=ad)
        feats.append(x)
        masks.append(mask)
    return feats, masks


def neck(p, cfg, feats, masks):
    """FPNIdentity.forward, necks.py:173-198."""
    if cfg['fpn_with_ln']:
        feats = [ln(p, 'neck.fpn_norms.%d.' % i, f) for i, f in enumerate(feats)]
    return feats, masks


def head_trunk(p, pre, cfg, x, mask, level=0):
    for i in range(cfg['head_num_layers'] - 1):
        x, _ = masked_conv1d(x, mask, p[pre + 'head.%d.conv.weight' % i], p.get(pre + 'head.%d.conv.bias' % i))
        if cfg['head_with_ln']:
            x = ln(p, pre + 'norm.%d.' % i, x)
        x = _relu(pre + 'norm.%d@%d' % (i, level), x)
    return x


def heads(p, cfg, feats, masks):
    """PtTransformerClsHead / RegHead forward (meta_archs.py:259-275, 334-349) + the permutes of :848-852."""
    cls, reg = [], []
    for l, (f, m) in enumerate(zip(feats, masks)):
        c, _ = masked_conv1d(head_trunk(p, 'cls_head.', cfg, f, m, l), m, p['cls_head.cls_head.conv.weight'],
                             p['cls_head.cls_head.conv.bias'])
        r, _ = masked_conv1d(head_trunk(p, 'reg_head.', cfg, f, m, l), m, p['reg_head.offset_head.conv.weight'],
                             p['reg_head.offset_head.conv.bias'])
        cls.append(c.permute(0, 2, 1))
        reg.append(F.relu(r * p['reg_head.scale.%d.scale' % l]).permute(0, 2, 1))
    return cls, reg


def points(cfg, level_lens, dtype=torch.float32):
    """PointGenerator buffers (loc_generators.py:52-92): [T_l,4] = (t, lo, hi, stride)."""
    out = []
    strides = [cfg['scale_factor'] ** i for i in range(cfg['fpn_start_level'], cfg['backbone_arch'][-1] + 1)]
    max_len = cfg['max_seq_len'] * cfg['max_buffer_len_factor']
    for n, s, rr in zip(level_lens, strides, cfg['regression_range']):
        t = torch.arange(0, max_len, s)[:, None].float()
        k = t.shape[0]
        pts = torch.cat((t, torch.as_tensor(rr, dtype=torch.float)[None].repeat(k, 1),
                         torch.as_tensor(s, dtype=torch.float)[None].repeat(k, 1)), dim=1)
        out.append(pts[:n].to(dtype))
    return out


def label_points_single(p, cfg, pts, seg, lab):
    """label_points_single_video, meta_archs.py:1253-1344."""
    tc = cfg['train_cfg']
    n = pts.shape[0]
    t, stride = pts[:, 0, None], pts[:, 3, None]
    lens = (seg[:, 1] - seg[:, 0])[None, :].repeat(n, 1)
    left, right = t - seg[None, :, 0], seg[None, :, 1] - t
    rel = ((right - left) / 2.0) / (stride * lens)

    def gauss(mu, sig):
        mu, sig = p[mu][lab].permute(1, 0), p[sig][lab].permute(1, 0)
        return (-(rel - mu) ** 2 / (2 * sig ** 2)).exp()
    g_cls, g_l, g_r = gauss('mu', 'sigma'), gauss('mu_reg_left', 'sigma_reg_left'), gauss('mu_reg_right', 'sigma_reg_right')
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
    ncls = p['mu'].shape[0]
    cls_t = (sel @ F.one_hot(lab, ncls).to(reg.dtype)).clamp(min=0.0, max=1.0)
    rows = torch.arange(n)
    return cls_t.detach(), (reg[rows, idx] / stride).detach(), g_cls[rows, idx], g_l[rows, idx], g_r[rows, idx]


def sigmoid_focal(x, t, alpha=0.25, gamma=2.0):
    """losses.py:5-52."""
    pr = torch.sigmoid(x)
    ce = F.binary_cross_entropy_with_logits(x, t, reduction="none")
    p_t = pr * t + (1 - pr) * (1 - t)
    return (alpha * t + (1 - alpha) * (1 - t)) * ce * ((1 - p_t) ** gamma)


def diou_1d(pred, tgt, eps=1e-8):
    """ctr_diou_loss_1d, losses.py:109-168."""
    lp, rp, lg, rg = pred[:, 0], pred[:, 1], tgt[:, 0], tgt[:, 1]
    inter = torch.min(rp, rg) + torch.min(lp, lg)
    iou = inter / ((lp + rp) + (lg + rg) - inter).clamp(min=eps)
    rho = 0.5 * (rp - lp - rg + lg)
    return 1.0 - iou + torch.square(rho / (torch.max(lp, lg) + torch.max(rp, rg)).clamp(min=eps))


def losses(p, cfg, masks, cls_logits, offsets, segments, labels, loss_normalizer, reduce_sim=None, n_known=0):
    """PtTransformer.losses (meta_archs.py:1374-1524, no CL distillation) -> (dict, new loss_normalizer)."""
    tc = cfg['train_cfg']
    level_lens = [c.shape[1] for c in cls_logits]
    pts = torch.cat(points(cfg, level_lens, cls_logits[0].dtype), dim=0)
    lab = [label_points_single(p, cfg, pts, s, l) for s, l in zip(segments, labels)]
    gt_cls = torch.stack([x[0] for x in lab])
    gt_off = torch.stack([x[1] for x in lab])
    w_cls, w_l, w_r = (torch.stack([x[i] for x in lab]) for i in (2, 3, 4))
    valid = torch.cat([m.squeeze(1) for m in masks], dim=1)
    pos = torch.logical_and(gt_cls.sum(-1) > 0, valid)
    num_pos = int(pos.sum().item())
    loss_normalizer = 0.9 * loss_normalizer + 0.1 * max(num_pos, 1)
    logits = torch.cat(cls_logits, dim=1)
    ls = tc['label_smoothing']
    target = gt_cls[valid] * (1 - ls) + ls / (cfg['num_classes'] + 1)
    w_cls = torch.where(pos, w_cls, torch.ones_like(w_cls))
    cls_loss = (sigmoid_focal(logits[valid], target).sum(-1) * w_cls[valid]).sum() / loss_normalizer
    if logits.shape[-1] != 1:
        sc = logits.masked_fill(valid.unsqueeze(-1) == False, -1e7)     # noqa: E712
        sc = torch.max(sc.softmax(-1), dim=1)[0]
        inv = torch.zeros_like(sc)
        for i, l in enumerate(labels):
            inv[i, l] = 1
        al_loss = (-inv * sc.log() - (1 - inv) * (1 - sc).log()).sum() / loss_normalizer
    else:
        al_loss = torch.zeros((1,))
    pred = torch.cat(offsets, dim=1)[pos]
    if num_pos == 0:
        reg_loss = 0 * pred.sum()
    else:
        reg_loss = (diou_1d(pred, gt_off[pos]) * ((w_l[pos] + w_r[pos]) / 2.0) * w_cls[pos]).sum() / loss_normalizer
    lw = tc['loss_weight'] if tc['loss_weight'] > 0 else cls_loss.detach() / max(reg_loss.item(), 0.01)
    final = cls_loss + reg_loss * lw + al_loss * tc['al_loss_weight']
    if n_known > 0 and cfg['cl_cfg']['name'] == 'l2p':
        final = final - 0.1 * reduce_sim
    return {'cls_loss': cls_loss, 'reg_loss': reg_loss, 'al_loss': al_loss, 'final_loss': final}, loss_normalizer


def prompt_forward(p, cfg, text_tm, prompt_idx):
    """Prompt.forward with a given index window (prompt.py:47-116, training path of meta_archs.py:761-767)."""
    def l2n(v):
        return v * torch.rsqrt(torch.maximum((v ** 2).sum(1, keepdim=True), torch.tensor(1e-12, dtype=v.dtype)))
    key_n, x_n = l2n(p['prompt.prompt_key']), l2n(text_tm.mean(dim=1))
    B = text_tm.shape[0]
    raw = p['prompt.prompt'][prompt_idx]
    batched = raw.reshape(B, -1, raw.shape[-1])
    reduce_sim = torch.sum(key_n[prompt_idx] * x_n.unsqueeze(1)) / B
    return torch.cat([batched, text_tm], dim=1), reduce_sim


def prompt_select(p, cfg, text_tm):
    """Prompt.forward's own selection (prompt_mask = None: inference, and training once the task window runs past the
    pool): top-k keys per sample by cosine similarity, then -- batchwise_prompt -- the top_k most frequent ids of the
    batch for every sample (prompt.py:66-82)."""
    def l2n(v):
        return v * torch.rsqrt(torch.maximum((v ** 2).sum(1, keepdim=True), torch.tensor(1e-12, dtype=v.dtype)))
    k, pool = cfg['cl_cfg']['topk'], cfg['cl_cfg']['pool_size']
    sim = l2n(text_tm.mean(dim=1)) @ l2n(p['prompt.prompt_key']).t()
    _, idx = torch.topk(sim, k=k, dim=1)
    ids, counts = torch.unique(idx, return_counts=True, sorted=True)
    if ids.shape[0] < pool:
        ids = torch.cat([ids, torch.full((pool - ids.shape[0],), int(idx.min()), dtype=ids.dtype)])
        counts = torch.cat([counts, torch.zeros(pool - counts.shape[0], dtype=counts.dtype)])
    _, major = torch.topk(counts, k=k)
    return ids[major].expand(text_tm.shape[0], -1)


def batch_inputs(cfg, video_list, training, dtype=torch.float32):
    """preprocessing + query_preprocessing, meta_archs.py:1134-1221 (padding to max_seq_len)."""
    feats = [v['feats'] for v in video_list if len(v['labels']) > 0]
    lens = torch.as_tensor([f.shape[-1] for f in feats])
    max_len = int(lens.max())
    if training or max_len <= cfg['max_seq_len']:
        max_len = cfg['max_seq_len']
    else:
        st = cfg['scale_factor'] ** cfg['backbone_arch'][-1]
        max_len = (max_len + st - 1) // st * st
    x = feats[0].new_zeros((len(feats), feats[0].shape[0], max_len))
    for f, d in zip(feats, x):
        d[..., :f.shape[-1]].copy_(f)
    mask = (torch.arange(max_len)[None, :] < lens[:, None]).unsqueeze(1)
    text = text_mask = None
    if cfg['use_cross_modal']:
        tf = [v['prompt_feature'] for v in video_list]
        tl = torch.as_tensor([f.shape[-1] for f in tf])
        text = tf[0].new_zeros((len(tf), tf[0].shape[0], int(tl.max())))
        for f, d in zip(tf, text):
            d[..., :f.shape[-1]].copy_(f)
        text_mask = (torch.arange(int(tl.max()))[None, :] < tl[:, None]).unsqueeze(1)
        text = text.to(dtype)
    return x.to(dtype), mask, text, text_mask


def forward_network(p, cfg, video_list, training=True, task_id=-1):
    dtype = p['mu'].dtype
    x, mask, text, text_mask = batch_inputs(cfg, video_list, training, dtype)
    reduce_sim = None
    adapter_blocks = tuple(cfg['cl_cfg']['adapt_blocks']) if cfg['cl_cfg'].get('use_adapt') else ()
    if 'prompt.prompt' in p:
        k = cfg['cl_cfg']['topk']
        if training and (task_id + 1) * k <= cfg['cl_cfg']['pool_size']:
            idx = torch.arange(task_id * k, (task_id + 1) * k).unsqueeze(0).expand(text.shape[0], -1)
        else:
            idx = prompt_select(p, cfg, text.permute(0, 2, 1))                 # meta_archs.py:766-769
        ttm, reduce_sim = prompt_forward(p, cfg, text.permute(0, 2, 1), idx)
        text = ttm.permute(0, 2, 1)
        tl = torch.as_tensor([v['prompt_feature'].shape[-1] for v in video_list])
        text_mask = (torch.arange(text.shape[-1])[None, :] < tl[:, None]).unsqueeze(1)   # meta_archs.py:775-779
    feats, masks = backbone(p, cfg, x, mask, text, text_mask, training, adapter_blocks)
    feats, masks = neck(p, cfg, feats, masks)
    cls, reg = heads(p, cfg, feats, masks)
    return feats, masks, cls, reg, reduce_sim


def forward_losses(p, cfg, video_list, loss_normalizer=None, task_id=-1, n_known=0):
    """model(video_list, is_training=True) of the reference in deterministic (eval-dropout) mode."""
    if loss_normalizer is None:
        loss_normalizer = cfg['train_cfg']['init_loss_norm']
    _, masks, cls, reg, reduce_sim = forward_network(p, cfg, video_list, True, task_id)
    segs = [v['segments'].to(p['mu'].dtype) for v in video_list if len(v['labels']) > 0]
    labs = [v['labels'] for v in video_list if len(v['labels']) > 0]
    return losses(p, cfg, masks, cls, reg, segs, labs, loss_normalizer, reduce_sim, n_known)


@torch.no_grad()
def decode_single_video(cfg, pts, masks, cls_logits, offsets):
    """inference_single_video, meta_archs.py:1594-1692 -> (segs, scores, labels) before NMS."""
    tc = cfg['test_cfg']
    segs, scores, labels = [], [], []
    for cls_i, off_i, pts_i, m_i in zip(cls_logits, offsets, pts, masks):
        prob = (cls_i.sigmoid() * m_i.unsqueeze(-1)).flatten()
        keep = prob > tc['pre_nms_thresh']
        prob, idx = prob[keep], keep.nonzero(as_tuple=True)[0]
        k = min(tc['pre_nms_topk'], idx.size(0))
        prob, order = prob.sort(descending=True)
        prob, idx = prob[:k].clone(), idx[order[:k]].clone()
        pt = torch.div(idx, cfg['num_classes'], rounding_mode='floor')
        offs, pp = off_i[pt], pts_i[pt]
        left, right = pp[:, 0] - offs[:, 0] * pp[:, 3], pp[:, 0] + offs[:, 1] * pp[:, 3]
        ok = (right - left) > tc['duration_thresh']
        segs.append(torch.stack((left, right), -1)[ok])
        scores.append(prob[ok])
        labels.append(torch.fmod(idx, cfg['num_classes'])[ok])
    return torch.cat(segs), torch.cat(scores), torch.cat(labels)


# ------------------------------------------------------------------------------ ViLCo extras
def ssl_embeddings(p, fpn_feats, fpn_masks, narration, narr_token_mask):
    """narration / video embeddings of the narration-SSL branch, meta_archs.py:794-811.
    fpn_feats: list of [B,C,T_l], fpn_masks: list of bool [B,1,T_l]; narration [B,Cn,n], narr_token_mask [B,1,n]."""
    nf = F.linear(narration.permute(0, 2, 1), p['narration_encoder.weight'], p['narration_encoder.bias']).permute(0, 2, 1)
    m1 = narr_token_mask.to(nf.dtype)
    den = m1.sum(dim=2)
    den = torch.where(den == 0, torch.ones_like(den), den)
    narr = F.normalize((nf * m1).sum(dim=2) / den, dim=1)
    pooled = []
    for f, m in zip(fpn_feats, fpn_masks):
        mf = m.to(f.dtype)
        d = mf.sum(dim=2)
        d = torch.where(d == 0, torch.ones_like(d), d)
        pooled.append((f * mf).sum(dim=2) / d)
    video = F.normalize(torch.stack(pooled).mean(dim=0), dim=1)
    return narr, video


def masked_contrastive_loss(text, video, mask, memory, temperature=0.07):
    """InfoNCE of each modality against the memory bank, positives first (meta_archs.py:1351-1372)."""
    t, v = text[mask], video[mask]
    pos = (t * v).sum(dim=1, keepdim=True)
    lt = torch.cat([pos, t @ memory.T], dim=1) / temperature
    lv = torch.cat([pos, v @ memory.T], dim=1) / temperature
    tgt = torch.zeros(t.shape[0], dtype=torch.long)
    return (F.cross_entropy(lt, tgt) + F.cross_entropy(lv, tgt)) / 2


def cl_penalty(named_params, importance_list, optpar_list, lam):
    """EWC.get_regularized_loss / MAS.get_mas_regularized_loss minus the base loss (EWC.py:6-22, MAS.py:5-21):
    lam * sum_tasks sum_names['scale' not in name] sum F (opt - p[:len(opt)])^2."""
    total = 0.0
    for imp_d, opt_d in zip(importance_list, optpar_list):
        for name, prm in named_params:
            if 'scale' not in name and name in imp_d:
                opt = opt_d[name]
                total = total + (imp_d[name] * (opt - prm[:opt.size(0)]).pow(2)).sum() * lam
    return total


def cl_distill(out_cls_logits, prev_out_cls_logits, n_known, cl_name, n_classes):
    """the distillation term iCaRL / BiC add to final_loss when n_known > 0 (meta_archs.py:1482-1519).
    out_cls_logits: list over pyramid levels of [B, T_l, ncls] logits; prev_out_cls_logits: what train_cl.py:226-235
    cached for the batch's clips -- a list over clips of lists over levels of [T_l, ncls_prev] sigmoid outputs (numpy in
    the reference).  Only clip 0 of the batch and the first cached clip enter (`out_cls_logits_i[0, ...]`,
    `prev_out_cls_logits[0]`): kept as is.
      bic  : 0.01 * n_known / n_classes * mean_t( -sum_c prev[t, c] * log_softmax(logits[0, t, :n_known] / 2)[c] ) per level
      icarl: 0.01 * sum_{y < n_known} BCEWithLogits(logits[0, :, y], prev[:, y]) per level"""
    len_f = len(out_cls_logits)
    prev = prev_out_cls_logits
    total = 0.0
    for i in range(len_f):
        cur = out_cls_logits[i]
        if cl_name == 'bic':
            pv = torch.as_tensor(prev[i]).to(cur.dtype)
            logp = F.log_softmax(cur[0, :, :n_known] / 2, dim=1)
            total = total + 0.01 * (n_known / n_classes) * -torch.mean(torch.sum(pv[:, :n_known] * logp, dim=1))
        else:
            if len(prev) != len_f or len(prev) == 1:          # a list over clips: take the first clip's levels (:1505-1506)
                prev = prev[0]
            pv = torch.as_tensor(prev[i]).to(cur.dtype)
            total = total + 0.01 * sum(F.binary_cross_entropy_with_logits(cur[0, :, y], pv[:, y]) for y in range(n_known))
    return total


@torch.no_grad()
def icarl_exemplar_means(p, cfg, memory, batches_of):
    """class means of the exemplars' pyramid features, meta_archs.py:1067-1096 (`classify`, compute_means branch).
    memory: {class_id: [clips]} in insertion order; batches_of(data_class) = cilsettask.get_dataloader(data_class,
    sample_frame=True) -> iterable of one-clip batches.  Every clip's level-l feature map [1, C, T_l] is divided by its
    Frobenius norm (padding included), averaged over the class's exemplars and normalised again.
    -> list over levels of [n_classes, C, T_l]."""
    means = None
    for class_id, videos in memory.items():
        per_level = None
        for video_list in batches_of({class_id: videos}):
            feats = forward_network(p, cfg, video_list, training=False)[0]
            f = [x / x.norm() for x in feats]
            per_level = [[a] for a in f] if per_level is None else [l + [a] for l, a in zip(per_level, f)]
        mus = []
        for lvl in per_level:
            mu = torch.stack(lvl, dim=0).mean(0).squeeze()            # [C, T_l]
            mus.append(mu / mu.norm())
        means = [[m] for m in mus] if means is None else [l + [m] for l, m in zip(means, mus)]
    return [torch.stack(l, dim=0) for l in means]


@torch.no_grad()
def icarl_dists(p, cfg, means, clip):
    """squared distance of every (level, position) feature of `clip` to every class mean, meta_archs.py:1098-1129:
    list over levels of [1, T_l, n_classes]."""
    feats = forward_network(p, cfg, [clip], training=False)[0]
    out = []
    for f, m in zip(feats, means):
        fn = (f / f.norm()).unsqueeze(3)                               # [1, C, T_l, 1]
        d = (fn - m.permute(1, 2, 0).unsqueeze(0)).pow(2).sum(1).squeeze()
        out.append(d.unsqueeze(0) if d.dim() == 2 else d)
    return out


@torch.no_grad()
def decode_single_video_icarl(cfg, pts, masks, cls_logits, offsets, dists):
    """inference_single_video with iCaRL's class distances (meta_archs.py:1626-1643, then the common tail :1663-1692):
    candidates are the (position, class) pairs closer to their class mean than the level's average distance, ranked by
    distance.  The rank indices address the UNFILTERED distance array but are applied to the filtered candidates (:1641-
    1642); when they would run past the end the reference keeps every candidate (:1637-1640).  Kept as is."""
    tc = cfg['test_cfg']
    segs, scores, labels = [], [], []
    for cls_i, off_i, pts_i, m_i, d_i in zip(cls_logits, offsets, pts, masks, dists):
        prob = (cls_i.sigmoid() * m_i.unsqueeze(-1)).flatten()
        d = d_i.flatten()
        keep = d < d.mean()
        prob, idx = prob[keep], keep.nonzero(as_tuple=True)[0]
        k = min(tc['pre_nms_topk'], idx.size(0))
        order = d.sort(descending=False)[1]
        if not order[:k].max() > prob.shape[0]:
            prob, idx = prob[order[:k]].clone(), idx[order[:k]].clone()
        pt = torch.div(idx, cfg['num_classes'], rounding_mode='floor')
        offs, pp = off_i[pt], pts_i[pt]
        left, right = pp[:, 0] - offs[:, 0] * pp[:, 3], pp[:, 0] + offs[:, 1] * pp[:, 3]
        ok = (right - left) > tc['duration_thresh']
        segs.append(torch.stack((left, right), -1)[ok])
        scores.append(prob[ok])
        labels.append(torch.fmod(idx, cfg['num_classes'])[ok])
    return torch.cat(segs), torch.cat(scores), torch.cat(labels)


def bic_correct(logits, splits, alphas, betas):
    """BiC bias layers on class-range slices of the logits (meta_archs.py:823-836); logits [..., ncls]."""
    parts, lo = [], 0
    for hi, a, b in zip(splits, alphas, betas):
        parts.append(a * logits[..., lo:hi] + b)
        lo = hi
    return torch.cat(parts, dim=-1)
