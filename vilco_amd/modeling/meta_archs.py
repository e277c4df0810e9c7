"""`LocPointTransformer` meta-architecture of the MQ task on the HIP path.

Reference: MQ/libs/modeling/meta_archs.py -- PtTransformerClsHead :183-275, PtTransformerRegHead
:278-349, PtTransformer :352-1736 (forward :753, preprocessing :1134, query_preprocessing :1184,
label_points_single_video :1253, losses :1374, inference_single_video :1594, postprocessing
:1695, Adapter :105-148, MemoryBank :38-60, BiasLayer :26-36).

Same constructor kwargs (`make_meta_arch('LocPointTransformer', **cfg['model'])`), same
`forward(video_list, ...)` contract (loss dict in training, list of per-video result dicts at
inference), same state_dict keys and the attributes / methods train_cl.py reads.  The dense work
(backbone, neck, heads) runs token-major on libvilco_hip.so; label assignment and the loss
reductions are [4536 x N] / [#valid x ncls] device tensor expressions under autograd.
"""
import math
import os

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops
from ..cl_methods import Prompt
from ..utils.nms import batched_nms
from ..utils.model_ema import ModelEmaV2
from .blocks import LayerNorm, MaskedConv1D, Scale, from_tm, lens_to_mask, to_tm
from .losses import ctr_diou_loss_1d, sigmoid_focal_loss
from .models import make_backbone, make_generator, make_neck, register_meta_arch


def normal_distribution(x, mu=0, sigma=1):
    return (-(x - mu) ** 2 / (2 * sigma ** 2)).exp()


def _as_device(x, device):
    """cached distillation targets: device tensors as train_cl.cache_prev_logits keeps them here, or the reference's
    numpy arrays (train_cl.py:226-235)"""
    return x.to(device) if torch.is_tensor(x) else torch.from_numpy(x).to(device)


class BiasLayer(nn.Module):
    """BiC bias-correction layer (meta_archs.py:26-36)."""

    def __init__(self):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, requires_grad=True))
        self.beta = nn.Parameter(torch.zeros(1, requires_grad=True))

    def forward(self, x):
        return self.alpha * x + self.beta

    def printParam(self, i):
        print(i, self.alpha.item(), self.beta.item())


class MemoryBank:
    """ring buffer of narration embeddings for the SSL loss (meta_archs.py:38-60)."""

    def __init__(self, size, feature_dim, device=None):
        self.size, self.feature_dim = size, feature_dim
        self.memory = torch.randn(size, feature_dim, device=device if device is not None else "cuda")
        self.ptr = 0

    @torch.no_grad()
    def update(self, features):
        n = features.size(0)
        assert n <= self.size, "Batch size must be less than or equal to memory bank size"
        end = self.ptr + n
        if end <= self.size:
            self.memory[self.ptr:end] = features
            self.ptr = end
        else:
            head = self.size - self.ptr
            self.memory[self.ptr:] = features[:head]
            self.memory[:end - self.size] = features[head:]
            self.ptr = end - self.size

    def get_all(self):
        return self.memory


class Adapter(nn.Module):
    """time-axis adapter of the ViLCo method (meta_archs.py:105-148): a Linear over the T axis,
    T -> 5T -> T/2, added in parallel to a stride-2 attention output."""

    def __init__(self, embed_dim, down_sample=5, mode="parallel", scale=None, act_layer=nn.GELU, stride=1):
        super().__init__()
        assert mode in ["before", "after", "parallel"], f"Unknown mode {mode}"
        hidden_dim = int(embed_dim * down_sample)
        self.layer = nn.Sequential(nn.Linear(embed_dim, hidden_dim), act_layer(),
                                   nn.Linear(hidden_dim, embed_dim // 2))
        self.mode = mode
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.layer[0].weight, a=math.sqrt(5))
        nn.init.zeros_(self.layer[0].bias)
        nn.init.zeros_(self.layer[2].weight)
        nn.init.zeros_(self.layer[2].bias)

    def forward_tm(self, x):
        """x [B,T,D] token-major -> [B,T/2,D]: the Linear acts on T, so run it channel-first."""
        if self.mode != "parallel":
            raise NotImplementedError("only the 'parallel' adapter mode is used (meta_archs.py:684)")
        xc = ops.transpose(x)                                           # [B,D,T]
        h = ops.linear(xc, self.layer[0].weight, self.layer[0].bias, ops.ACT_GELU)
        h = ops.linear(h, self.layer[2].weight, self.layer[2].bias)     # [B,D,T/2]
        return ops.transpose(h)


def freeze(module):
    for p in module.parameters():
        p.requires_grad_(False)
        p.grad = None


def unfreeze(module):
    for p in module.parameters():
        p.requires_grad_(True)


class _ConvHead(nn.Module):
    """[conv k3 -> LN -> ReLU] x (num_layers-1) shared across pyramid levels."""

    def _build(self, input_dim, feat_dim, num_layers, kernel_size, with_ln):
        self.head, self.norm = nn.ModuleList(), nn.ModuleList()
        for idx in range(num_layers - 1):
            self.head.append(MaskedConv1D(input_dim if idx == 0 else feat_dim, feat_dim, kernel_size,
                                          stride=1, padding=kernel_size // 2, bias=(not with_ln)))
            self.norm.append(LayerNorm(feat_dim) if with_ln else nn.Identity())

    def _trunk(self, x, lens):
        for conv, norm in zip(self.head, self.norm):
            x, _ = conv.forward_tm(x, lens)
            x = norm.forward_tm(x, relu=True) if isinstance(norm, LayerNorm) else torch.relu(x)
        return x

    # ---- all pyramid levels in one pass.  The head weights are shared across levels (meta_archs.py:216-235), so the
    # levels of each clip are laid end to end as ONE token sequence with a zero row between neighbours: a k=3 conv
    # over it equals the per-level convs (each level sees the zero padding it would get alone), the GEMMs get
    # M = B * sum(T_l) rows instead of six small problems, and every other kernel of the trunk runs once.
    # Rows beyond a level's valid length keep the reference's semantics (conv output * mask, then LN / ReLU of the
    # zero row); separator rows are forced back to zero after every layer.
    def _conv_cat(self, conv, x, cat):
        c = conv.conv
        assert c.kernel_size[0] == 3 and conv.stride == 1 and c.groups == 1
        if not _HEAD_ROWMASK:                        # (A/B: the mask as a separate multiply, as up to round 3)
            return ops.conv3(x, c.weight, c.bias, None) * cat.valid
        return ops.conv3(x, c.weight, c.bias, None, row_mask=cat.valid)          # output rows masked in the GEMM epilogue

    def _trunk_cat(self, x, cat):
        for conv, norm in zip(self.head, self.norm):
            x = self._conv_cat(conv, x, cat)
            if isinstance(norm, LayerNorm):
                # the separator rows are zeroed by the LayerNorm kernel itself, which also writes the next conv's operand image
                x = norm.forward_tm(x, relu=True, planes="seq", row_mask=cat.notgap)
            else:
                x = torch.relu(x)
        return x

    @staticmethod
    def can_cat(heads):
        return all(m.conv.kernel_size[0] == 3 and m.stride == 1 and m.conv.groups == 1 for m in heads)


_HEAD_ROWMASK = os.environ.get("VILCO_HEAD_ROWMASK", "1") != "0"


class LevelCat:
    """Pyramid levels [B,T_l,C] of every clip concatenated along T with one zero row between levels."""

    _layout = {}     # (T_l ..., device) -> constant index tensors of the concatenated layout

    @classmethod
    def _get_layout(cls, Ts, dev):
        key = (tuple(Ts), str(dev))
        lay = cls._layout.get(key)
        if lay is None:
            off, lvl, pos, o = [], [], [], 0
            for i, T in enumerate(Ts):
                if i:
                    lvl.append(i); pos.append(1 << 30); o += 1          # separator row: never valid
                off.append(o)
                lvl.extend([i] * T); pos.extend(range(T)); o += T
            notgap = torch.tensor([0.0 if p_ == (1 << 30) else 1.0 for p_ in pos], dtype=torch.float32, device=dev)
            lay = (off, o, torch.tensor(lvl, dtype=torch.long, device=dev), torch.tensor(pos, dtype=torch.int32, device=dev),
                   notgap[None, :, None])
            cls._layout = {key: lay}
        return lay

    def __init__(self, feats, lens):
        B, dev = feats[0].shape[0], feats[0].device
        self.T = [f.shape[1] for f in feats]
        self.off, total, lvl, pos, self.notgap = self._get_layout(self.T, dev)
        zero = feats[0].new_zeros(B, 1, feats[0].shape[2])
        pieces = []
        for i, f in enumerate(feats):
            if i:
                pieces.append(zero)
            pieces.append(f)
        self.x = torch.cat(pieces, dim=1)
        lim = torch.stack([l.to(torch.int32) for l in lens], dim=1)[:, lvl]            # [B,Tc] valid length of the row's level
        self.valid = (pos[None, :] < lim).to(torch.float32)[:, :, None]                 # [B,Tc,1]

    def split(self, y):
        return [y[:, o:o + T] for o, T in zip(self.off, self.T)]


class PtTransformerClsHead(_ConvHead):
    def __init__(self, input_dim, feat_dim, num_classes, prior_prob=0.01, num_layers=3, kernel_size=3,
                 act_layer=nn.ReLU, with_ln=False, empty_cls=[], detach_feat=False):
        super().__init__()
        self.act = act_layer()
        self.detach_feat, self.num_classes = detach_feat, num_classes
        self._build(input_dim, feat_dim, num_layers, kernel_size, with_ln)
        self.cls_head = MaskedConv1D(feat_dim, num_classes, kernel_size, stride=1, padding=kernel_size // 2)
        torch.nn.init.constant_(self.cls_head.conv.bias, -(math.log((1 - prior_prob) / prior_prob)))
        if len(empty_cls) > 0:
            for idx in empty_cls:
                torch.nn.init.constant_(self.cls_head.conv.bias[idx], -(math.log((1 - 1e-6) / 1e-6)))
        self.reg_params = {}

    def augment_classification(self, num_new_classes, device):
        self.cls_head.augment_classification(num_new_classes, device)
        self.num_classes += num_new_classes

    def forward_tm(self, feats, lens, cat=None):
        """-> list of logits [B, T_l, ncls] (already the permuted layout of meta_archs.py:848)."""
        if cat is not None and self.can_cat(list(self.head) + [self.cls_head]):
            x = self._trunk_cat(cat.x.detach() if self.detach_feat else cat.x, cat)
            cat.cls_logits = self._conv_cat(self.cls_head, x, cat)       # [B, Tc, ncls]: the fused loss reads it whole
            return cat.split(cat.cls_logits)
        out = []
        for x, l in zip(feats, lens):
            x = self._trunk(x.detach() if self.detach_feat else x, l)
            out.append(self.cls_head.forward_tm(x, l)[0])
        return out

    def forward(self, fpn_feats, fpn_masks):
        lens = [m.reshape(m.shape[0], -1).sum(1).to(torch.int32) for m in fpn_masks]
        return tuple(from_tm(y) for y in self.forward_tm([to_tm(f) for f in fpn_feats], lens))


class PtTransformerRegHead(_ConvHead):
    def __init__(self, input_dim, feat_dim, fpn_levels, num_layers=3, kernel_size=3, act_layer=nn.ReLU,
                 with_ln=False, num_bins=16):
        super().__init__()
        self.fpn_levels = fpn_levels
        self.act = act_layer()
        self._build(input_dim, feat_dim, num_layers, kernel_size, with_ln)
        self.scale = nn.ModuleList([Scale() for _ in range(fpn_levels)])
        self.offset_head = MaskedConv1D(feat_dim, 2 * (num_bins + 1), kernel_size, stride=1,
                                        padding=kernel_size // 2)
        self.reg_params = {}

    def forward_tm(self, feats, lens, cat=None, raw=False):
        """raw=True (training with the fused loss kernel, which applies relu(Scale_l(x)) itself): only `cat.raw_offsets`
        [B, Tc, 2] is produced and None returned."""
        assert len(feats) == self.fpn_levels
        if cat is not None and self.can_cat(list(self.head) + [self.offset_head]):
            cat.raw_offsets = self._conv_cat(self.offset_head, self._trunk_cat(cat.x, cat), cat)
            if raw:
                return None
            off = cat.split(cat.raw_offsets)
            return [F.relu(self.scale[l](o)) for l, o in enumerate(off)]
        out = []
        for l, (x, ln) in enumerate(zip(feats, lens)):
            x = self._trunk(x, ln)
            off, _ = self.offset_head.forward_tm(x, ln)
            out.append(F.relu(self.scale[l](off)))       # [B, T_l, 2]: tiny, plain device ops
        return out

    def forward(self, fpn_feats, fpn_masks):
        lens = [m.reshape(m.shape[0], -1).sum(1).to(torch.int32) for m in fpn_masks]
        return tuple(from_tm(y) for y in self.forward_tm([to_tm(f) for f in fpn_feats], lens))


class StepInputs:
    """device-resident inputs of one step (PtTransformer.prepare): feats_cf [B,Cin,T], lens int32 [B], text_cf
    [B,Ctxt,L], text_lens int32 [B], gt float [B, 3*Nmax+1] (training), narr = narration tuple or None"""
    __slots__ = ("feats_cf", "lens", "T", "text_cf", "text_lens", "narr", "gt")

    def tensors(self):
        """(name, tensor) of every device buffer a replayed step reads"""
        return [(k, getattr(self, k)) for k in ("feats_cf", "lens", "text_cf", "text_lens", "gt") if getattr(self, k) is not None]

    def signature(self):
        return tuple((k, tuple(t.shape)) for k, t in self.tensors()) + (("narr", self.narr is not None),)


@register_meta_arch("LocPointTransformer")
class PtTransformer(nn.Module):
    def __init__(self, backbone_type, fpn_type, use_xl, backbone_arch, scale_factor, input_dim,
                 max_seq_len, max_buffer_len_factor, n_head, n_mha_win_size, embd_kernel_size, embd_dim,
                 embd_with_ln, fpn_dim, fpn_with_ln, fpn_start_level, head_dim, regression_range,
                 head_num_layers, head_kernel_size, head_with_ln, use_abs_pe, use_rel_pe, num_classes,
                 train_cfg, test_cfg, cl_cfg, use_cross_modal, n_txt_in, xlnet_config=None):
        super().__init__()
        self.fpn_strides = [scale_factor ** i for i in range(fpn_start_level, backbone_arch[-1] + 1)]
        self.reg_range = regression_range
        assert len(self.fpn_strides) == len(self.reg_range)
        self.scale_factor, self.num_classes, self.max_seq_len = scale_factor, num_classes, max_seq_len
        if isinstance(n_mha_win_size, int):
            self.mha_win_size = [n_mha_win_size] * (1 + backbone_arch[-1])
        else:
            assert len(n_mha_win_size) == (1 + backbone_arch[-1])
            self.mha_win_size = n_mha_win_size
        max_div_factor = 1
        for s, w in zip(self.fpn_strides, self.mha_win_size):
            stride = s * (w // 2) * 2 if w > 1 else s
            assert max_seq_len % stride == 0, "max_seq_len must be divisible by fpn stride and window size"
            max_div_factor = max(max_div_factor, stride)
        self.max_div_factor, self.use_xl = max_div_factor, use_xl

        t = train_cfg
        self.train_center_sample = t['center_sample']
        assert self.train_center_sample in ['radius', 'none']
        self.train_center_sample_radius = t['center_sample_radius']
        self.train_loss_weight, self.train_cls_prior_prob = t['loss_weight'], t['cls_prior_prob']
        self.train_dropout, self.train_droppath = t['dropout'], t['droppath']
        self.train_label_smoothing = t['label_smoothing']
        self.t_c_alpha, self.al_loss_weight = t['t_c_alpha'], t['al_loss_weight']
        self.cont_loss_weight, self.seg_loss_weight = t['cont_loss_weight'], t['seg_loss_weight']
        self.queue_size, self.temperature = t['queue_size'], t['temperature']
        self.use_dcn, self.dcn_start_layer, self.use_us_fpn = t['use_dcn'], t['dcn_start_layer'], t['use_us_fpn']
        self.length_theta, self.num_bins, self.iou_weight_power = t['length_theta'], t["num_bins"], t["iou_weight_power"]

        self.test_pre_nms_thresh, self.test_pre_nms_topk = test_cfg['pre_nms_thresh'], test_cfg['pre_nms_topk']
        self.test_iou_threshold, self.test_min_score = test_cfg['iou_threshold'], test_cfg['min_score']
        self.test_max_seg_num, self.test_nms_method = test_cfg['max_seg_num'], test_cfg['nms_method']
        assert self.test_nms_method in ['soft', 'hard', 'none']
        self.test_duration_thresh, self.test_multiclass_nms = test_cfg['duration_thresh'], test_cfg['multiclass_nms']
        self.test_nms_sigma, self.test_voting_thresh = test_cfg['nms_sigma'], test_cfg['voting_thresh']
        self.use_cross_modal, self.n_txt_in = use_cross_modal, n_txt_in

        assert backbone_type in ['convTransformer', 'conv']
        if backbone_type != 'convTransformer':
            raise NotImplementedError("backbone_type 'conv' is not used by any shipped MQ config")
        bb = {'n_in': input_dim, 'n_embd': embd_dim, 'n_head': n_head, 'n_embd_ks': embd_kernel_size,
              'max_len': max_seq_len, 'use_xl': use_xl, 'arch': backbone_arch, 't_c_alpha': self.t_c_alpha,
              'scale_factor': scale_factor, 'with_ln': embd_with_ln, 'attn_pdrop': 0.0,
              'proj_pdrop': self.train_dropout, 'path_pdrop': self.train_droppath, 'use_abs_pe': use_abs_pe,
              'use_rel_pe': use_rel_pe, 'use_dcn': self.use_dcn, 'dcn_start_layer': self.dcn_start_layer,
              'use_cross_modal': self.use_cross_modal, 'n_txt_in': self.n_txt_in}
        if xlnet_config is not None:
            bb['xlnet_config'] = xlnet_config
        self.backbone = make_backbone('convTransformer', **bb)
        if isinstance(embd_dim, (list, tuple)):
            embd_dim = sum(embd_dim)
        assert fpn_type in ['fpn', 'identity']
        self.neck = make_neck(fpn_type, **{'in_channels': [embd_dim] * (backbone_arch[-1] + 1),
                                           'out_channel': fpn_dim, 'scale_factor': scale_factor,
                                           'start_level': fpn_start_level, 'with_ln': fpn_with_ln,
                                           'use_us_fpn': self.use_us_fpn})
        self.point_generator = make_generator('point', **{'max_seq_len': max_seq_len * max_buffer_len_factor,
                                                           'fpn_strides': self.fpn_strides,
                                                           'regression_range': self.reg_range,
                                                           'use_us_fpn': self.use_us_fpn})
        self.cls_head = PtTransformerClsHead(fpn_dim, head_dim, self.num_classes, kernel_size=head_kernel_size,
                                             prior_prob=self.train_cls_prior_prob, with_ln=head_with_ln,
                                             num_layers=head_num_layers, empty_cls=t['head_empty_cls'])
        self.reg_head = PtTransformerRegHead(fpn_dim, head_dim, len(self.fpn_strides),
                                             kernel_size=head_kernel_size, num_layers=head_num_layers,
                                             with_ln=head_with_ln, num_bins=0)

        # run both heads over all pyramid levels at once (LevelCat); VILCO_LEVEL_CAT=0 keeps the per-level loop
        self.level_cat = os.environ.get("VILCO_LEVEL_CAT", "1") != "0"

        nc = self.num_classes
        self.mu = nn.Parameter(torch.zeros(nc, 1), requires_grad=True)
        self.sigma = nn.Parameter(torch.ones(nc, 1), requires_grad=True)
        self.mu_reg_left = nn.Parameter(-torch.ones(nc, 1) * 0.5, requires_grad=True)
        self.sigma_reg_left = nn.Parameter(torch.ones(nc, 1), requires_grad=True)
        self.mu_reg_right = nn.Parameter(torch.ones(nc, 1) * 0.5, requires_grad=True)
        self.sigma_reg_right = nn.Parameter(torch.ones(nc, 1), requires_grad=True)

        self.loss_normalizer = t['init_loss_norm']     # EMA, not checkpointed (:611); a device scalar in the sync-free path
        self.sync_free_loss = os.environ.get("VILCO_SYNC_FREE_LOSS", "1") != "0"
        self.fused_loss = os.environ.get("VILCO_FUSED_LOSS", "1") != "0"      # ops.mq_loss (0: tensor expressions)
        self.loss_normalizer_momentum = 0.9
        self.reg_params = {}

        self.compute_means = cl_cfg['name'] == 'icarl'
        self.exemplar_means, self.memory = [], {}
        self.adv_lambda, self.type_sampling = cl_cfg['adv_lambda'], cl_cfg['type_sampling']
        self.n_known = 0
        self.dist_loss = nn.BCEWithLogitsLoss()
        self.list_bias_layers, self.list_splits = [], []
        self.cl_name = cl_cfg['name']

        self.prompt_pool = cl_cfg['prompt_pool']
        self.use_prompt_mask = True
        if cl_cfg['length'] is not None and cl_cfg['pool_size'] is not None and self.prompt_pool:
            self.prompt = Prompt(length=cl_cfg['length'], embed_dim=cl_cfg['embed_dim'], embedding_key='mean',
                                 prompt_init='uniform', prompt_pool=True, prompt_key=True,
                                 pool_size=cl_cfg['pool_size'], top_k=cl_cfg['topk'], batchwise_prompt=True,
                                 prompt_key_init='uniform')
        self.narration_ssl, self.narration_dim = cl_cfg["narration_ssl"], cl_cfg["narration_dim"]
        if self.narration_ssl:
            self.narration_encoder = nn.Linear(cl_cfg['narration_dim'], 1024)
            self._memory_bank_cfg = (cl_cfg['memory_size'], 1024)
            self.memory_bank = None        # created on first use, on the model's device
        self.ssl_factor = cl_cfg["ssl_factor"]

        self.num_emas, self.ema_decay = 1, 0.999
        self.use_adapt = cl_cfg['use_adapt']
        if self.use_adapt:
            self.adapt_blocks = cl_cfg['adapt_blocks']
            self.num_freeze_epochs = 10
            self.setup_adpat()

    # ------------------------------------------------------------------ adapters / EMA (ViLCo)
    def setup_adpat(self):
        if getattr(self, "pets_emas", None) is None:
            self.pets_emas = nn.ModuleList([])
            self.pets = self.create_pets()
        if len(self.pets_emas) < self.num_emas:
            self.pets_emas.append(ModelEmaV2(self.pets, decay=self.ema_decay))
        self.attach_pets(self.pets)

    def attach_pets(self, pets):
        for i, b in enumerate(self.adapt_blocks):
            self.backbone.branch[b].attach_adapter(attn=pets[i])

    def create_pets(self):
        pets, dim = nn.ModuleList([]), 1024          # hard-wired to T=1024 in the reference (:682)
        for _ in range(len(self.adapt_blocks)):
            pets.append(Adapter(embed_dim=dim, down_sample=5, mode='parallel', scale='null'))
            dim //= 2
        return pets

    def pre_train_epoch(self, task_id=0, current_epoch=0):
        unfreeze(self.pets)

    def post_train_step(self):
        for idx, ema in enumerate(reversed(self.pets_emas)):
            ema.update(self.pets if idx == 0 else self.pets_emas[idx - 1])

    @property
    def device(self):
        """the reference walks every parameter here (meta_archs.py:709-713; ~1 ms of host time per call at 465 tensors,
        several calls per step); all parameters live on one device, so the first one answers"""
        return self.mu.device

    def augment_classification(self, num_new_classes, device):
        device = self.mu.device
        self.cls_head.augment_classification(num_new_classes, device)
        old = self.num_classes
        self.num_classes += num_new_classes
        inits = {'mu': 0.0, 'sigma': 1.0, 'mu_reg_left': -0.5, 'sigma_reg_left': 1.0,
                 'mu_reg_right': 0.5, 'sigma_reg_right': 1.0}
        for name, val in inits.items():
            new = nn.Parameter(torch.full((self.num_classes, 1), val, device=device), requires_grad=True)
            new.data[:old] = getattr(self, name).data
            setattr(self, name, new)

    # ------------------------------------------------------------------ batching
    def _batch_cf(self, video_list, is_training=True, padding_val=0.0):
        """list of dicts -> (batched [B,C,T] channel-first on device, lens int32 [B], max_len)  (meta_archs.py:1134-1181)"""
        feats = [x['feats'] for x in video_list if len(x['labels']) > 0]
        feats_lens = [f.shape[-1] for f in feats]
        max_len = max(feats_lens)
        if is_training:
            assert max_len <= self.max_seq_len, "Input length must be smaller than max_seq_len during training"
            max_len = self.max_seq_len
        else:
            assert len(video_list) == 1, "Only support batch_size = 1 during inference"
            if max_len <= self.max_seq_len:
                max_len = self.max_seq_len
            else:
                stride = self.max_div_factor
                max_len = (max_len + (stride - 1)) // stride * stride
        dev = self.device
        batched = torch.full((len(feats), feats[0].shape[0], max_len), padding_val, dtype=torch.float32, device=dev)
        for f, dst in zip(feats, batched):
            dst[..., :f.shape[-1]].copy_(f, non_blocking=True)       # H2D (or D2D) of the raw [C, t_i]
        lens = self._h2d(torch.as_tensor(feats_lens, dtype=torch.int32))
        return batched, lens, max_len

    def preprocessing(self, video_list, is_training=True, padding_val=0.0):
        """list of dicts -> (x_tm [B,T,C] on device, lens int32 [B], max_len)  (meta_archs.py:1134-1181;
        the reference returns channel-first [B,C,T] + bool mask: see `preprocessing_cf`)."""
        batched, lens, max_len = self._batch_cf(video_list, is_training, padding_val)
        return ops.transpose(batched), lens, max_len

    def preprocessing_cf(self, video_list, is_training=True, padding_val=0.0):
        """reference-shaped return: (batched_inputs [B,C,T], batched_masks [B,1,T] bool, None)."""
        x_tm, lens, T = self.preprocessing(video_list, is_training, padding_val)
        return ops.transpose(x_tm), lens_to_mask(lens, T), None

    @torch.no_grad()
    def query_preprocessing(self, video_list, padding_val=0.0):
        """text tokens [768, L_i] -> token-major [B, Lmax, 768] + lens (meta_archs.py:1184-1221)."""
        batched, lens, narr = self._query_batch_cf(video_list, padding_val)
        return ops.transpose(batched), lens, narr

    @torch.no_grad()
    def _query_batch_cf(self, video_list, padding_val=0.0):
        """-> (batched [B,768,Lmax] channel-first on device, lens int32 [B], narration tuple or None)"""
        feats = [x['prompt_feature'] for x in video_list]
        lens_h = [f.shape[-1] for f in feats]
        dev = self.device
        batched = torch.full((len(feats), feats[0].shape[0], max(lens_h)), padding_val, dtype=torch.float32, device=dev)
        for f, dst in zip(feats, batched):
            dst[..., :f.shape[-1]].copy_(f, non_blocking=True)
        lens = self._h2d(torch.as_tensor(lens_h, dtype=torch.int32))
        narr = None
        if self.training and self.narration_ssl:
            nf = [x['narration_feats'] for x in video_list]
            nl = [f.shape[-1] for f in nf]
            nb = torch.full((len(nf), nf[0].shape[0], max(nl)), padding_val, dtype=torch.float32, device=dev)
            for f, dst in zip(nf, nb):
                dst[..., :f.shape[-1]].copy_(f, non_blocking=True)
            m0 = torch.Tensor([x['narration_mask'] for x in video_list]).to(dev)
            m1 = (torch.arange(max(nl))[None, :] < torch.as_tensor(nl)[:, None]).unsqueeze(1).to(dev)
            narr = (nb, m0, m1)
        return batched, lens, narr

    def _gt_table(self, video_list, nmax=None):
        """ground truth of the labelled clips as ONE float table [B, 3*Nmax+1] for the fused label / loss kernels
        (include/vilco_hip.h: vilco_loss_desc.gt): 2*Nmax segment bounds, Nmax labels, the count.  Nmax may be padded
        (the kernels stop at the count), which keeps the table's shape stable from batch to batch."""
        vids = [x for x in video_list if len(x['labels']) > 0]
        n_real = max(int(x['labels'].shape[0]) for x in vids)
        nmax = n_real if nmax is None else max(int(nmax), n_real)
        gt = torch.zeros(len(vids), 3 * nmax + 1, dtype=torch.float32)
        for b, x in enumerate(vids):
            n = int(x['labels'].shape[0])
            gt[b, :2 * n] = x['segments'].reshape(-1).float().cpu()
            gt[b, 2 * nmax:2 * nmax + n] = x['labels'].float().cpu()
            gt[b, 3 * nmax] = n
        return self._h2d(gt)

    def _h2d(self, host):
        """small host table -> device without stalling the host: a copy from pageable memory waits for everything queued
        on the stream before it (one drained queue per step = the next step's ~1.6 ms of launches exposed, measured with
        tools/lab/replay_ab.py at config P); from a pinned staging block it is just another queued copy"""
        dev = self.device
        if dev.type == 'cuda':
            host = host.pin_memory()
        return host.to(dev, non_blocking=True)

    def prepare(self, video_list, is_training=True, gt_pad=None):
        """Everything `forward` takes from the clip dictionaries, as device tensors: the host half of the step (padding,
        H2D copies, the ground-truth table).  `forward_prepared` is the device half -- no host reads, no H2D copies, so a
        training step over a `StepInputs` can be captured as a hipGraph and replayed over refreshed buffers
        (vilco_amd/graph.py).  The channel-first -> token-major layout change is part of the device half."""
        inp = StepInputs()
        inp.feats_cf, inp.lens, inp.T = self._batch_cf(video_list, is_training)
        inp.text_cf = inp.text_lens = inp.narr = None
        if self.use_cross_modal:
            inp.text_cf, inp.text_lens, inp.narr = self._query_batch_cf(video_list)
            assert inp.text_cf.shape[0] == inp.feats_cf.shape[0], \
                "every clip of the batch needs labels (the reference drops unlabelled clips from the video batch only)"
        inp.gt = None
        if is_training and video_list[0].get('segments') is not None and video_list[0].get('labels') is not None:
            inp.gt = self._gt_table(video_list, gt_pad)
        return inp

    # ------------------------------------------------------------------ forward
    def _run_network(self, x_tm, lens, text_tm, text_lens, raw_offsets=False):
        feats, all_lens = self.backbone.forward_tm(x_tm, lens, text_tm, text_lens)
        fpn_feats, fpn_lens = self.neck.forward_tm(feats, all_lens)
        cat = LevelCat(fpn_feats, fpn_lens) if (self.level_cat and len(fpn_feats) > 1) else None
        self._cat = cat
        if ops.fork_enabled("heads"):
            # the two heads share nothing but their input: the regression trunk runs on a side stream beside the
            # classification trunk (each of their k=3 convs is 1.5 rounds of tiles on 256 CUs: together 3 instead of 4)
            main, side = torch.cuda.current_stream(), ops.side_stream("heads", fpn_feats[0].device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                out_offsets = self.reg_head.forward_tm(fpn_feats, fpn_lens, cat, raw=raw_offsets and cat is not None)
            out_cls_logits = self.cls_head.forward_tm(fpn_feats, fpn_lens, cat)
            main.wait_stream(side)
        else:
            out_offsets = self.reg_head.forward_tm(fpn_feats, fpn_lens, cat, raw=raw_offsets and cat is not None)
            out_cls_logits = self.cls_head.forward_tm(fpn_feats, fpn_lens, cat)
        return fpn_feats, fpn_lens, out_cls_logits, out_offsets

    def forward(self, video_list, task_id=-1, ensemble=False, hidden_state=False, is_training=True,
                prev_out_cls_logits=None, get_emb=False, val_qilDatasetList=None):
        inp = self.prepare(video_list, is_training)
        return self.forward_prepared(inp, video_list, task_id=task_id, ensemble=ensemble, is_training=is_training,
                                     prev_out_cls_logits=prev_out_cls_logits, get_emb=get_emb,
                                     val_qilDatasetList=val_qilDatasetList)

    def capturable(self, inp, task_id=-1, prev_out_cls_logits=None):
        """can forward_prepared(inp, None, task_id) + backward run without touching the host?  (the fused label / loss
        kernels, a fixed prompt window, no narration SSL -- whose memory-bank update reads a device flag --, no
        distillation against host-side logits)"""
        if not (self.fused_loss and self.sync_free_loss and self.train_loss_weight > 0 and self.num_classes <= 128):
            return False
        if inp.gt is None or (self.training and self.narration_ssl) or prev_out_cls_logits:
            return False
        if self.n_known > 0 and self.cl_name in ('bic', 'icarl'):
            return False
        if hasattr(self, 'prompt') and not (0 <= task_id and (task_id + 1) * self.prompt.top_k <= self.prompt.pool_size):
            return False
        return True

    def forward_prepared(self, inp, video_list=None, task_id=-1, ensemble=False, is_training=True,
                         prev_out_cls_logits=None, get_emb=False, val_qilDatasetList=None):
        """the device half of `forward` over a StepInputs; `video_list` is only needed by the paths that read the clip
        dictionaries (inference meta data, the unfused label path)."""
        x_tm, lens = ops.transpose(inp.feats_cf), inp.lens
        text_tm = text_lens = None
        narr = inp.narr
        if self.use_cross_modal:
            text_tm, text_lens = ops.transpose(inp.text_cf), inp.text_lens

        reduce_sim = None
        if hasattr(self, 'prompt'):
            # L2P prompts are prepended to the text tokens (meta_archs.py:759-780).  The reference
            # rebuilds the mask over the first len(text) positions of the prompted sequence only.
            prompt_mask = None
            if is_training:
                start, end = task_id * self.prompt.top_k, (task_id + 1) * self.prompt.top_k
                if end <= self.prompt.pool_size:
                    prompt_mask = torch.arange(start, end, device=text_tm.device).unsqueeze(0).expand(text_tm.shape[0], -1)
            res = self.prompt(text_tm, prompt_mask=prompt_mask, cls_features=None)
            self.total_prompt_len = res['total_prompt_len']
            text_tm = res['prompted_embedding'].contiguous()
            reduce_sim = res['reduce_sim']

        # training with fixed loss weights: labels + losses are ONE fused kernel pair (ops.mq_loss) that also applies the
        # regression head's relu(Scale_l(x)); everything else (BiC, dynamic loss weight, get_emb, eval) keeps the lists
        fused = (is_training and not get_emb and self.fused_loss and self.sync_free_loss and self.train_loss_weight > 0
                 and not (self.n_known > 0 and self.cl_name == 'bic') and self.num_classes <= 128)
        fpn_feats, fpn_lens, out_cls_logits, out_offsets = self._run_network(x_tm, lens, text_tm, text_lens,
                                                                            raw_offsets=fused)

        if self.training and self.narration_ssl:
            narration_feats, video_feats = self._ssl_embeddings(fpn_feats, fpn_lens, narr)

        level_T = [f.shape[1] for f in fpn_feats]
        points = self.point_generator(fpn_feats, lengths=level_T)

        if self.n_known > 0 and self.cl_name == 'bic':
            out_cls_logits = [self._bic_correct(x) for x in out_cls_logits]

        # [B, T_l] bool per level -- not needed by the fused training losses (they read the prefix lengths): 12 launches less on the
        # chain of a captured step (round 6)
        fpn_masks = (None if (is_training and fused and not get_emb)
                     else [lens_to_mask(l, T).squeeze(1) for l, T in zip(fpn_lens, level_T)])

        if not is_training and self.use_adapt:
            # average with the prediction of every adapter EMA (meta_archs.py:854-881)
            for ema in self.pets_emas:
                self.attach_pets(ema.module)
                _, _, e_cls, e_off = self._run_network(x_tm, lens, text_tm, text_lens)
                out_cls_logits = [(a + b) / 2 for a, b in zip(out_cls_logits, e_cls)]
                out_offsets = [(a + b) / 2 for a, b in zip(out_offsets, e_off)]
            self.attach_pets(self.pets)

        if get_emb:
            return out_cls_logits, out_offsets, fpn_masks

        if is_training:
            assert inp.gt is not None, "GT action labels does not exist"
            dev = self.device
            if fused:
                losses = self._fused_losses(inp.gt, points, fpn_lens, out_cls_logits, out_offsets,
                                            prev_out_cls_logits, reduce_sim)
            else:
                gt_segments = [x['segments'].to(dev) for x in video_list if len(x['labels']) > 0]
                gt_labels = [x['labels'].to(dev) for x in video_list if len(x['labels']) > 0]
                gt_cls, gt_off, np_cls, np_reg = self.label_points(points, gt_segments, gt_labels)
                losses = self.losses(fpn_masks, out_cls_logits, out_offsets, gt_cls, gt_off, label_list=gt_labels,
                                     normal_probs_cls=np_cls, normal_probs_reg=np_reg,
                                     prev_out_cls_logits=prev_out_cls_logits, reduce_sim=reduce_sim)
            if self.narration_ssl and narr[1].sum() > 0:
                m0 = narr[1].to(torch.bool)
                self.memory_bank.update(narration_feats[m0])
                ssl_loss = self.masked_contrastive_loss(narration_feats, video_feats, m0)
                losses["final_loss"] += self.ssl_factor * ssl_loss
                losses["ssl_loss"] = self.ssl_factor * ssl_loss
            return losses

        results = self.inference(video_list, points, fpn_masks, out_cls_logits, out_offsets, None, None,
                                 val_qilDatasetList)
        if ensemble:
            return video_list, points, fpn_masks, out_cls_logits, out_offsets
        return results

    def _bic_correct(self, logits):
        parts, lo = [], 0
        for i, hi in enumerate(self.list_splits):
            parts.append(self.list_bias_layers[i](logits[..., lo:hi]))
            lo = hi
        return torch.cat(parts, dim=-1)

    def _ssl_embeddings(self, fpn_feats, fpn_lens, narr):
        """masked mean pooling of narration / pyramid features (meta_archs.py:794-811)."""
        nb, m0, m1 = narr
        if self.memory_bank is None:
            self.memory_bank = MemoryBank(*self._memory_bank_cfg, device=self.device)
        nf = self.narration_encoder(nb.permute(0, 2, 1)).permute(0, 2, 1) * m1
        denom = torch.sum(m1, dim=2, dtype=torch.float)
        denom[denom == 0.] = 1.
        narration_feats = F.normalize(torch.sum(nf, dim=2) / denom, dim=1)
        pooled = []
        for f, l in zip(fpn_feats, fpn_lens):
            m = lens_to_mask(l, f.shape[1]).squeeze(1).to(f.dtype)             # [B,T]
            d = m.sum(1, keepdim=True)
            d[d == 0.] = 1.
            pooled.append((f * m.unsqueeze(-1)).sum(1) / d)
        video_feats = F.normalize(torch.stack(pooled).mean(0), dim=1)
        return narration_feats, video_feats

    def masked_contrastive_loss(self, text_embeddings, video_embeddings, mask, temperature=0.07):
        t, v = text_embeddings[mask], video_embeddings[mask]
        pos = torch.einsum('nc,nc->n', [t, v]).unsqueeze(-1)
        mem = self.memory_bank.get_all()
        lt = torch.cat([pos, t @ mem.T], dim=1) / temperature
        lv = torch.cat([pos, v @ mem.T], dim=1) / temperature
        labels = torch.zeros(t.size(0), dtype=torch.long, device=t.device)
        return (F.cross_entropy(lt, labels) + F.cross_entropy(lv, labels)) / 2

    # ------------------------------------------------------------------ labels
    def label_points(self, points, gt_segments, gt_labels, for_seg=False):
        concat_points = torch.cat(points, dim=0)
        gt_cls, gt_offset, np_cls, np_reg = [], [], [], []
        for seg, lab in zip(gt_segments, gt_labels):
            cls_t, reg_t, (p_cls, p_l, p_r) = self.label_points_single_video(concat_points, seg, lab)
            gt_cls.append(cls_t.detach())
            gt_offset.append(reg_t.detach())
            np_cls.append(p_cls)
            np_reg.append([p_l, p_r])
        return gt_cls, gt_offset, np_cls, np_reg

    def label_points_single_video(self, concat_points, gt_segment, gt_label):
        """all-pairs point/GT assignment with gaussian weights from the learnable mu/sigma
        (meta_archs.py:1253-1344)."""
        num_pts, num_gts = concat_points.shape[0], gt_segment.shape[0]
        if num_gts == 0:
            return gt_segment.new_full((num_pts, self.num_classes), 0), gt_segment.new_zeros((num_pts, 2))
        t = concat_points[:, 0, None]
        stride = concat_points[:, 3, None]
        lens = (gt_segment[:, 1] - gt_segment[:, 0])[None, :].repeat(num_pts, 1)
        left = t - gt_segment[None, :, 0]
        right = gt_segment[None, :, 1] - t
        rel = ((right - left) / 2.0) / (stride * lens)

        def gauss(mu, sigma):
            return normal_distribution(rel, mu[gt_label].permute(1, 0), sigma[gt_label].permute(1, 0))
        p_cls = gauss(self.mu, self.sigma)
        p_left = gauss(self.mu_reg_left, self.sigma_reg_left)
        p_right = gauss(self.mu_reg_right, self.sigma_reg_right)
        reg_targets = torch.stack((left, right), dim=-1)

        if self.train_center_sample == 'radius':
            center = 0.5 * (gt_segment[None, :, 0] + gt_segment[None, :, 1])
            t_mins = center - stride * self.train_center_sample_radius
            t_maxs = center + stride * self.train_center_sample_radius
            cb_left = t - torch.maximum(t_mins, gt_segment[None, :, 0])
            cb_right = torch.minimum(t_maxs, gt_segment[None, :, 1]) - t
            inside = torch.minimum(cb_left, cb_right) > 0
        else:
            inside = reg_targets.min(-1)[0] > 0
        max_dist = reg_targets.max(-1)[0]
        in_range = torch.logical_and(max_dist >= concat_points[:, 1, None], max_dist <= concat_points[:, 2, None])
        lens = lens.masked_fill(inside == 0, float('inf')).masked_fill(in_range == 0, float('inf'))
        min_len, min_len_inds = lens.min(dim=1)
        min_len_mask = torch.logical_and(lens <= (min_len[:, None] + 1e-3), lens < float('inf')).to(reg_targets.dtype)
        one_hot = F.one_hot(gt_label, self.num_classes).to(reg_targets.dtype)
        cls_targets = (min_len_mask @ one_hot).clamp_(min=0.0, max=1.0)
        rows = torch.arange(num_pts, device=concat_points.device)
        reg_targets = reg_targets[rows, min_len_inds] / stride
        return cls_targets, reg_targets, (p_cls[rows, min_len_inds], p_left[rows, min_len_inds],
                                          p_right[rows, min_len_inds])

    # ------------------------------------------------------------------ fused labels + losses
    def _loss_tables(self, points, cat, dev):
        """(points [R,4], row_level [R], row_pos [R]) of the row layout the loss kernel walks: the LevelCat layout with
        separator rows (stride 0) or the plain concatenation of the levels.  Cached per layout."""
        key = (tuple(p.shape[0] for p in points), cat is not None, str(dev))
        tab = getattr(self, "_loss_tab", None)
        if tab is None or tab[0] != key:
            rows, lvl, pos = [], [], []
            for i, p in enumerate(points):
                if i and cat is not None:
                    rows.append(p.new_zeros(1, 4))
                    lvl.append(i)
                    pos.append(1 << 30)
                rows.append(p)
                lvl.extend([i] * p.shape[0])
                pos.extend(range(p.shape[0]))
            tab = (key, (torch.cat(rows).contiguous().to(dev), torch.tensor(lvl, dtype=torch.int32, device=dev),
                         torch.tensor(pos, dtype=torch.int32, device=dev)))
            self._loss_tab = tab
        return tab[1]

    def _fused_losses(self, gt, points, fpn_lens, out_cls_logits, out_offsets, prev_out_cls_logits, reduce_sim):
        """meta_archs.py:1253-1344 + 1374-1447 through ops.mq_loss; the CL terms (:1478-1519) are added on top.
        gt: the table of `_gt_table` -- one row per clip of the batch, in batch order."""
        cat, self._cat = self._cat, None       # not kept: it holds this step's head tensors and their autograd graph
        dev = self.device
        if cat is not None and getattr(cat, "raw_offsets", None) is not None and out_offsets is None:
            logits, offsets = cat.cls_logits, cat.raw_offsets
            scale = torch.stack([s.scale for s in self.reg_head.scale])
        else:
            logits, offsets, scale, cat = torch.cat(out_cls_logits, dim=1), torch.cat(out_offsets, dim=1), None, None
        assert gt.shape[0] == logits.shape[0], "ground-truth rows (%d) != clips in the batch (%d)" % (gt.shape[0], logits.shape[0])
        tables = self._loss_tables(points, cat, dev)
        level_len = torch.stack([l.to(torch.int32) for l in fpn_lens], dim=1).contiguous()
        gauss = torch.cat([self.mu, self.sigma, self.mu_reg_left, self.sigma_reg_left, self.mu_reg_right,
                           self.sigma_reg_right], dim=1).t().contiguous()                         # [6, ncls]
        norm = self._loss_norm
        if not (torch.is_tensor(norm) and norm.dim() == 1):
            norm = torch.full((1,), float(norm), dtype=torch.float32, device=dev)
            self._loss_norm = norm
        radius = self.train_center_sample_radius if self.train_center_sample == 'radius' else 0.0
        cls_loss, reg_loss, al_loss, final_loss = ops.mq_loss(
            logits, offsets, scale, gauss, tables, level_len, gt, norm, radius, self.train_label_smoothing,
            self.loss_normalizer_momentum, self.train_loss_weight, self.al_loss_weight, logits.shape[-1] != 1)
        return self._cl_terms({'cls_loss': cls_loss, 'reg_loss': reg_loss, 'al_loss': al_loss}, final_loss,
                              out_cls_logits, prev_out_cls_logits, reduce_sim)

    # ------------------------------------------------------------------ losses
    @property
    def loss_normalizer(self):
        """python float like the reference's attribute; reading it after a sync-free step waits for the device."""
        v = self._loss_norm
        return float(v) if torch.is_tensor(v) else v

    @loss_normalizer.setter
    def loss_normalizer(self, v):
        self._loss_norm = float(v)

    def _loss_norm_tensor(self, device):
        v = self._loss_norm
        return v.reshape(()) if torch.is_tensor(v) else torch.tensor(float(v), dtype=torch.float32, device=device)

    def losses(self, fpn_masks, out_cls_logits, out_offsets, gt_cls_labels, gt_offsets, label_list=None,
               normal_probs_cls=None, normal_probs_reg=None, out_importances=None, out_start=None,
               out_end=None, prev_out_cls_logits=None, stage_id=0, reduce_sim=None):
        """focal + DIoU with gaussian point weights, 'al' loss, CL distillation terms
        (meta_archs.py:1374-1524).  One host sync: num_pos (as in the reference, :1407)."""
        valid_mask = torch.cat(fpn_masks, dim=1)
        gt_cls = torch.stack(gt_cls_labels)
        w_cls = torch.stack(normal_probs_cls)
        w_left = torch.stack([x[0] for x in normal_probs_reg])
        w_right = torch.stack([x[1] for x in normal_probs_reg])
        pos_mask = torch.logical_and((gt_cls.sum(-1) > 0), valid_mask)
        logits_all = torch.cat(out_cls_logits, dim=1)
        offsets_all = torch.cat(out_offsets, dim=1)
        gt_off_all = torch.stack(gt_offsets)
        gt_target = gt_cls * (1 - self.train_label_smoothing) + self.train_label_smoothing / (self.num_classes + 1)
        w_cls = torch.where(pos_mask, w_cls, torch.ones_like(w_cls))           # negatives weigh 1

        if self.sync_free_loss:
            # Same sums as the reference, evaluated densely over all points with 0/1 weights instead of boolean
            # gathers: no host round trip anywhere in the step (the reference syncs on num_pos, on every boolean
            # index and in the DIoU asserts), so the GPU never waits for the host between forward and backward.
            num_pos = pos_mask.sum()
            norm = self._loss_norm_tensor(logits_all.device)
            norm = self.loss_normalizer_momentum * norm + (1 - self.loss_normalizer_momentum) * num_pos.clamp(min=1).to(norm.dtype)
            self._loss_norm = norm.detach()
            vf, pf = valid_mask.to(logits_all.dtype), pos_mask.to(logits_all.dtype)
            cls_loss = sigmoid_focal_loss(logits_all, gt_target, reduction='None')
            cls_loss = (cls_loss.sum(-1) * w_cls * vf).sum() / norm
            one = torch.ones_like(gt_off_all)
            reg = ctr_diou_loss_1d(torch.where(pos_mask[..., None], offsets_all, one).reshape(-1, 2),
                                   torch.where(pos_mask[..., None], gt_off_all, one).reshape(-1, 2),
                                   reduction='None', check=False).reshape(pos_mask.shape)
            reg_loss = (reg * ((w_left + w_right) / 2.0) * w_cls * pf).sum() / norm
        else:
            pred_offsets = offsets_all[pos_mask]
            gt_off = gt_off_all[pos_mask]
            num_pos = pos_mask.sum().item()
            self.loss_normalizer = self.loss_normalizer_momentum * self.loss_normalizer + (
                1 - self.loss_normalizer_momentum) * max(num_pos, 1)
            norm = self.loss_normalizer
            cls_loss = sigmoid_focal_loss(logits_all[valid_mask], gt_target[valid_mask], reduction='None')
            cls_loss = (cls_loss.sum(-1) * w_cls[valid_mask]).sum() / norm
            if num_pos == 0:
                reg_loss = 0 * pred_offsets.sum()
            else:
                reg_loss = ctr_diou_loss_1d(pred_offsets, gt_off, reduction='None')
                reg_loss = reg_loss * ((w_left[pos_mask] + w_right[pos_mask]) / 2.0) * w_cls[pos_mask]
                reg_loss = reg_loss.sum() / norm

        if label_list is not None and (out_cls_logits[0].shape[-1] != 1):
            score = logits_all.masked_fill(valid_mask.unsqueeze(-1) == False, -1e7)   # noqa: E712
            score = torch.max(score.softmax(-1), dim=1)[0]
            involved = torch.zeros_like(score)
            for i in range(involved.shape[0]):
                involved[i, label_list[i]] = 1
            al_loss = (-involved * score.log() - (1 - involved) * (1 - score).log()).sum() / norm
        else:
            al_loss = torch.zeros((1,), device=cls_loss.device)

        if self.train_loss_weight > 0:
            loss_weight = self.train_loss_weight
        else:
            loss_weight = cls_loss.detach() / max(reg_loss.item(), 0.01)
        final_loss = cls_loss + reg_loss * loss_weight + al_loss * self.al_loss_weight

        return self._cl_terms({'cls_loss': cls_loss, 'reg_loss': reg_loss, 'al_loss': al_loss}, final_loss, out_cls_logits,
                              prev_out_cls_logits, reduce_sim)

    def _cl_terms(self, out, final_loss, out_cls_logits, prev_out_cls_logits, reduce_sim):
        """continual-learning terms on top of the detection loss (meta_archs.py:1478-1519)"""
        if self.n_known > 0 and self.cl_name == 'l2p':
            final_loss = final_loss - 0.1 * reduce_sim
        if self.n_known > 0 and self.cl_name == 'bic':
            n_classes = self.cls_head.cls_head.conv.out_channels
            alpha, temp, dist_loss = self.n_known / n_classes, 2, 0
            for cur, prev in zip(out_cls_logits, prev_out_cls_logits):
                prev = _as_device(prev, cur.device)
                logp = F.log_softmax(cur[0, :, :self.n_known] / temp, dim=1)
                dist_loss = dist_loss + 0.01 * alpha * -torch.mean(torch.sum(prev[:, :self.n_known] * logp, dim=1))
            final_loss = final_loss + dist_loss
            out['dist_loss'] = dist_loss
        if self.n_known > 0 and self.cl_name == 'icarl':
            len_f, dist_loss = len(out_cls_logits), 0
            for i in range(len_f):
                if len(prev_out_cls_logits) != len_f or len(prev_out_cls_logits) == 1:
                    prev_out_cls_logits = prev_out_cls_logits[0]
                prev = _as_device(prev_out_cls_logits[i], out_cls_logits[i].device)
                dist_loss = dist_loss + 0.01 * sum(self.dist_loss(out_cls_logits[i][0, :, y], prev[:, y])
                                                   for y in range(self.n_known))
            final_loss = final_loss + dist_loss
            out['dist_loss'] = dist_loss
        out['final_loss'] = final_loss
        return out

    # ------------------------------------------------------------------ iCaRL nearest-exemplar-mean classification
    @torch.no_grad()
    def _pyramid_features(self, video_list):
        """backbone + neck of an inference batch (one clip): list over levels of token-major [1, T_l, C].
        The text goes in RAW: the reference's classify hands query_preprocessing's tokens and mask straight to the backbone
        (meta_archs.py:1077-1079, 1108-1110), also on a model with an L2P prompt pool."""
        inp = self.prepare(video_list, is_training=False)
        x_tm = ops.transpose(inp.feats_cf)
        text_tm = text_lens = None
        if self.use_cross_modal:
            text_tm, text_lens = ops.transpose(inp.text_cf), inp.text_lens
        feats, all_lens = self.backbone.forward_tm(x_tm, inp.lens, text_tm, text_lens)
        return self.neck.forward_tm(feats, all_lens)[0]

    @torch.no_grad()
    def classify(self, x, cilsettask):
        """iCaRL's classifier (meta_archs.py:1061-1131): squared distances between the clip's normalised pyramid features
        and the class means of the exemplars in `self.memory` -> list over levels of [1, T_l, n_classes].
        The means are (re)computed while `compute_means` is set (`cilsettask.get_dataloader({class: clips},
        sample_frame=True)` yields the exemplars as one-clip batches): each exemplar's level-l map is divided by its
        Frobenius norm, the maps are averaged per class and normalised again.
        One difference: the reference hard-codes `fpn_levels = 10` (:1065) and therefore fails on any other pyramid
        depth (the shipped (2, 2, 5) architectures have 6 levels); the level count is taken from the model here.  On a
        10-level model the outputs equal the reference's (tests/test_icarl.py)."""
        if self.compute_means:
            means = None
            for class_id, videos in self.memory.items():
                per_level = None
                for video_list in cilsettask.get_dataloader({class_id: videos}, sample_frame=True):
                    f = [t / t.norm() for t in self._pyramid_features(video_list)]
                    per_level = [[a] for a in f] if per_level is None else [l + [a] for l, a in zip(per_level, f)]
                mus = []
                for lvl in per_level:
                    mu = torch.stack(lvl, dim=0).mean(0)[0]               # [T_l, C]
                    mus.append(mu / mu.norm())
                means = [[m] for m in mus] if means is None else [l + [m] for l, m in zip(means, mus)]
            self.exemplar_means = [torch.stack(l, dim=0) for l in means]          # per level [n_classes, T_l, C]
            self.compute_means = False
        dists = []
        for f, m in zip(self._pyramid_features([x]), self.exemplar_means):
            fn = f / f.norm()                                                      # [1, T_l, C]
            dists.append((fn.unsqueeze(1) - m.unsqueeze(0)).pow(2).sum(-1).permute(0, 2, 1).contiguous())   # [1, T_l, n_cls]
        return dists

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def inference(self, video_list, points, fpn_masks, out_cls_logits, out_offsets, out_lb_logits,
                  out_rb_logits, cilsettask=None):
        results = []
        for idx, vl in enumerate(video_list):
            # final validation of an iCaRL run (meta_archs.py:1561-1562): class distances of this clip.  `classify`
            # clears compute_means once the exemplar means exist, so -- as in the reference -- only the first clip
            # validated after training is decoded from distances; the following ones take the ordinary path.
            cls_preds = self.classify(vl, cilsettask) if (cilsettask is not None and self.compute_means) else None
            r = self.inference_single_video(points, [x[idx] for x in fpn_masks],
                                            [x[idx] for x in out_cls_logits], [x[idx] for x in out_offsets],
                                            None, None, cls_preds_per_vid=cls_preds)
            for k_out, k_in in (('video_id', 'video_id'), ('fps', 'fps'), ('duration', 'duration'),
                                ('feat_stride', 'feat_stride'), ('feat_num_frames', 'feat_num_frames')):
                r[k_out] = vl[k_in]
            results.append(r)
        return self.postprocessing(results)

    @torch.no_grad()
    def inference_single_video(self, points, fpn_masks, out_cls_logits, out_offsets, lb_logits_per_vid=None,
                               rb_logits_per_vid=None, candidate_label=None, cls_preds_per_vid=None):
        """threshold -> top-k -> decode per level (meta_archs.py:1594-1692).  On the device the whole pyramid of the clip
        is one vilco_decode call (exact top-k by radix select, one host read: the candidate count); VILCO_DEVICE_DECODE=0
        or host tensors take the tensor-expression path below, which mirrors the reference line by line."""
        if cls_preds_per_vid is None and out_cls_logits[0].is_cuda and os.environ.get("VILCO_DEVICE_DECODE", "1") != "0":
            lens = [int(c.shape[0]) for c in out_cls_logits]
            row0 = torch.tensor([sum(lens[:i]) for i in range(len(lens))], dtype=torch.int32, device=out_cls_logits[0].device)
            level_len = torch.stack([m.sum() for m in fpn_masks]).to(torch.int32)
            segs, scores, labels = ops.decode(torch.cat(out_cls_logits).float().contiguous(), torch.cat(out_offsets).float().contiguous(),
                                              torch.cat(points).float().contiguous(), row0, level_len, self.test_pre_nms_topk,
                                              self.test_pre_nms_thresh, self.test_duration_thresh)
            return {'segments': segs, 'scores': scores, 'labels': labels}
        segs_all, scores_all, cls_all = [], [], []
        for lvl, (cls_i, off_i, pts_i, mask_i) in enumerate(zip(out_cls_logits, out_offsets, points, fpn_masks)):
            prob = (cls_i.sigmoid() * mask_i.unsqueeze(-1)).flatten()
            if cls_preds_per_vid is not None:
                # iCaRL (meta_archs.py:1626-1643): candidates = (position, class) pairs closer to their class mean than the
                # level's average distance, ranked by distance.  The rank indices address the unfiltered distance array
                # but are applied to the filtered candidates; when they would run past the end every candidate is kept.
                dist = cls_preds_per_vid[lvl].flatten()
                keep = dist < dist.mean()
                prob = prob[keep]
                topk_idxs = keep.nonzero(as_tuple=True)[0]
                num_topk = min(self.test_pre_nms_topk, topk_idxs.size(0))
                order = dist.sort(descending=False)[1]
                if not bool(order[:num_topk].max() > prob.shape[0]):
                    prob = prob[order[:num_topk]].clone()
                    topk_idxs = topk_idxs[order[:num_topk]].clone()
            else:
                keep = prob > self.test_pre_nms_thresh
                prob = prob[keep]
                topk_idxs = keep.nonzero(as_tuple=True)[0]
                num_topk = min(self.test_pre_nms_topk, topk_idxs.size(0))
                prob, order = prob.sort(descending=True)
                prob = prob[:num_topk].clone()
                topk_idxs = topk_idxs[order[:num_topk]].clone()
            pt_idxs = torch.div(topk_idxs, self.num_classes, rounding_mode='floor')
            cls_idxs = torch.fmod(topk_idxs, self.num_classes)
            offs, pts = off_i[pt_idxs], pts_i[pt_idxs]
            left = pts[:, 0] - offs[:, 0] * pts[:, 3]
            right = pts[:, 0] + offs[:, 1] * pts[:, 3]
            keep2 = (right - left) > self.test_duration_thresh
            segs_all.append(torch.stack((left, right), -1)[keep2])
            scores_all.append(prob[keep2])
            cls_all.append(cls_idxs[keep2])
        return {'segments': torch.cat(segs_all), 'scores': torch.cat(scores_all), 'labels': torch.cat(cls_all)}

    @torch.no_grad()
    def postprocessing(self, results):
        """device NMS, then seconds conversion and the single D2H copy (meta_archs.py:1695-1736)."""
        out = []
        for r in results:
            segs, scores, labels = r['segments'].detach(), r['scores'].detach(), r['labels'].detach()
            if self.test_nms_method != 'none':
                segs, scores, labels = batched_nms(segs, scores, labels, self.test_iou_threshold,
                                                   self.test_min_score, self.test_max_seg_num,
                                                   use_soft_nms=(self.test_nms_method == 'soft'),
                                                   multiclass=self.test_multiclass_nms, sigma=self.test_nms_sigma,
                                                   voting_thresh=self.test_voting_thresh)
            if segs.shape[0] > 0:
                segs = (segs * r['feat_stride'] + 0.5 * r['feat_num_frames']) / r['fps']
                segs = torch.where(segs <= 0.0, segs * 0.0, segs)
                vlen = r['duration']
                segs = torch.where(segs >= vlen, segs * 0.0 + vlen, segs)
            out.append({'video_id': r['video_id'], 'segments': segs.cpu(), 'scores': scores.cpu(),
                        'labels': labels.cpu()})
        return out

    # ------------------------------------------------------------------ replay memory (train_cl.py:343-361)
    def add_samples_to_mem(self, cilsettask, data, m):
        """random replay-memory sampling: merge the episode's {class: videos} into the memory, shuffle
        each class in place and keep m per class ('ALL' keeps everything) (meta_archs.py:1046-1052)."""
        import random
        self.memory = {**self.memory, **data}
        for class_id, videos in self.memory.items():
            random.shuffle(videos)
            self.memory[class_id] = videos[:m] if m != 'ALL' else videos
        for class_id, videos in self.memory.items():
            print('Memory... Class: {}, num videos: {}'.format(class_id, len(videos)))
