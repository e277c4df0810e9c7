"""`LocPointTransformer` meta-architecture of the NLQ task (natural-language moment queries) on the HIP path.

Reference: NLQ/libs/modeling/meta_archs.py -- PtTransformerClsHead :182-262, PtTransformerRegHead :265-337,
PtTransformer :340-1382 (forward :614, query_preprocessing :879, preprocessing :919, label_points_single_video :981,
losses :1094, inference_single_video :1253, postprocessing :1341).

The NLQ model is the MQ model (vilco_amd/modeling/meta_archs.py) minus the learnable gaussian point weights and the
action-localisation loss, with its own two-stream backbone (vilco_amd/modeling_nlq/backbones.py: the text query
goes through the backbone's cross-attention, video self-attention is a sliding window), one-hot labels in the batch
dicts and one query class.  This class therefore reuses the MQ implementation -- heads over all pyramid levels at
once (LevelCat), the fused label + loss kernels (vilco_mq_loss_*), decode + device NMS -- and restates only what
differs: the constructor (the reference's kwargs and state_dict keys: no mu / sigma parameters), batching of the
`feats` / `query_feats` / `one_hot_labels` dicts, and the loss configuration.  The gaussian weights of the fused loss
kernel are switched off by constant (non-persistent) buffers with sigma = 1e15: exp(-(rel - mu)^2 / 2 sigma^2) is
exactly 1.0f.
"""
import os

import torch
from torch import nn

from ..cl_methods import Prompt
from ..modeling import meta_archs as mq
from ..modeling.models import make_generator, make_neck
from .models import make_backbone, register_meta_arch


@register_meta_arch("LocPointTransformer")
class PtTransformer(mq.PtTransformer):
    def __init__(self, backbone_type, fpn_type, backbone_arch, scale_factor, input_vid_dim, input_txt_dim, max_seq_len,
                 max_buffer_len_factor, n_head, n_mha_win_size, embd_kernel_size, embd_dim, embd_with_ln, fpn_dim,
                 fpn_with_ln, fpn_start_level, head_dim, regression_range, head_num_layers, head_kernel_size,
                 head_with_ln, use_abs_pe, use_rel_pe, num_classes, train_cfg, test_cfg, cl_cfg):
        nn.Module.__init__(self)
        self.input_txt_dim = input_txt_dim
        n_levels = backbone_arch[-2] + backbone_arch[-1] + 1
        self.fpn_strides = [scale_factor ** i for i in range(fpn_start_level, n_levels)]
        self.reg_range = regression_range
        assert len(self.fpn_strides) == len(self.reg_range), (self.fpn_strides, self.reg_range)
        self.scale_factor, self.num_classes, self.max_seq_len = scale_factor, num_classes, max_seq_len
        if isinstance(n_mha_win_size, int):
            self.mha_win_size = [n_mha_win_size] * n_levels
        else:
            assert len(n_mha_win_size) == n_levels
            self.mha_win_size = n_mha_win_size
        max_div_factor = 1
        for s, w in zip(self.fpn_strides, self.mha_win_size):
            stride = s * (w // 2) * 2 if w > 1 else s
            assert max_seq_len % stride == 0, "max_seq_len %d must be divisible by fpn stride and window size %d" % (
                max_seq_len, stride)
            max_div_factor = max(max_div_factor, stride)
        self.max_div_factor, self.use_xl = max_div_factor, False

        t = train_cfg
        self.train_center_sample = t['center_sample']
        assert self.train_center_sample in ['radius', 'none']
        self.train_center_sample_radius = t['center_sample_radius']
        self.train_loss_weight, self.train_cls_prior_prob = t['loss_weight'], t['cls_prior_prob']
        self.train_dropout, self.train_droppath = t['dropout'], t['droppath']
        self.train_label_smoothing = t['label_smoothing']
        self.al_loss_weight = 0.0                                   # no action-localisation term in the NLQ losses (:1094-1198)

        self.test_pre_nms_thresh, self.test_pre_nms_topk = test_cfg['pre_nms_thresh'], test_cfg['pre_nms_topk']
        self.test_iou_threshold, self.test_min_score = test_cfg['iou_threshold'], test_cfg['min_score']
        self.test_max_seg_num, self.test_nms_method = test_cfg['max_seg_num'], test_cfg['nms_method']
        assert self.test_nms_method in ['soft', 'hard', 'none']
        self.test_duration_thresh, self.test_multiclass_nms = test_cfg['duration_thresh'], test_cfg['multiclass_nms']
        self.test_nms_sigma, self.test_voting_thresh = test_cfg['nms_sigma'], test_cfg['voting_thresh']
        self.use_cross_modal, self.n_txt_in = True, input_txt_dim
        use_adapter = cl_cfg['use_adapter']

        assert backbone_type == 'convTransformer'
        self.backbone = make_backbone('convTransformer', **{
            'n_vid_in': input_vid_dim, 'n_txt_in': input_txt_dim, 'n_embd': embd_dim, 'n_head': n_head,
            'n_embd_ks': embd_kernel_size, 'max_len': max_seq_len, 'arch': backbone_arch,
            'mha_win_size': self.mha_win_size, 'scale_factor': scale_factor, 'with_ln': embd_with_ln,
            'attn_pdrop': 0.0, 'proj_pdrop': self.train_dropout, 'path_pdrop': self.train_droppath,
            'use_abs_pe': use_abs_pe, 'use_rel_pe': use_rel_pe, 'use_adapter': use_adapter})
        assert fpn_type == 'identity'
        self.neck = make_neck(fpn_type, **{'in_channels': [embd_dim] * n_levels, 'out_channel': fpn_dim,
                                           'scale_factor': scale_factor, 'start_level': fpn_start_level,
                                           'with_ln': fpn_with_ln})
        self.point_generator = make_generator('point', **{'max_seq_len': max_seq_len * max_buffer_len_factor,
                                                           'fpn_strides': self.fpn_strides,
                                                           'regression_range': self.reg_range})
        assert self.num_classes > 0
        self.cls_head = mq.PtTransformerClsHead(fpn_dim, head_dim, self.num_classes, kernel_size=head_kernel_size,
                                                prior_prob=self.train_cls_prior_prob, with_ln=head_with_ln,
                                                num_layers=head_num_layers, empty_cls=t['head_empty_cls'])
        self.reg_head = mq.PtTransformerRegHead(fpn_dim, head_dim, len(self.fpn_strides), kernel_size=head_kernel_size,
                                                num_layers=head_num_layers, with_ln=head_with_ln, num_bins=0)
        self.level_cat = os.environ.get("VILCO_LEVEL_CAT", "1") != "0"

        # the MQ loss kernels weight every point by gaussians of learnable (mu, sigma); NLQ has no such weights:
        # constant buffers (not parameters, not in the state_dict) with sigma = 1e15 make every weight exactly 1
        nc = self.num_classes
        for name, val in (('mu', 0.0), ('sigma', 1e15), ('mu_reg_left', 0.0), ('sigma_reg_left', 1e15),
                          ('mu_reg_right', 0.0), ('sigma_reg_right', 1e15)):
            self.register_buffer(name, torch.full((nc, 1), val), persistent=False)

        self.loss_normalizer = t['init_loss_norm']
        self.sync_free_loss = os.environ.get("VILCO_SYNC_FREE_LOSS", "1") != "0"
        self.fused_loss = os.environ.get("VILCO_FUSED_LOSS", "1") != "0"
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
            self.narration_encoder = nn.Linear(cl_cfg['narration_dim'], 384)        # hard-wired feature_dim (:551)
            self._memory_bank_cfg = (cl_cfg['memory_size'], 384)
            self.memory_bank = None
        self.ssl_factor = cl_cfg["ssl_factor"]

        self.num_emas, self.ema_decay = 1, 0.999
        self.use_adapt = self.use_adapter = use_adapter
        if self.use_adapter:
            self.adapt_blocks = cl_cfg['adapt_blocks']
            self.num_freeze_epochs = 10
            self.setup_adpat()

    def augment_classification(self, num_new_classes, device):
        raise NotImplementedError("the NLQ reference has no class-head growth (queries are one class)")

    # ------------------------------------------------------------------ batching (:879-957)
    @staticmethod
    def _with_labels(video_list):
        """class indices for the MQ code paths out of NLQ's one-hot rows (meta_archs.py:1052: cls_targets = mask @ one_hot)"""
        out = []
        for x in video_list:
            oh = x['one_hot_labels']
            assert oh.dim() == 2, "one_hot_labels: [#segments, #classes]"
            if oh.numel() and not bool(((oh.sum(-1) == 1) & ((oh == 0) | (oh == 1)).all(-1)).all()):
                raise NotImplementedError("multi-hot query labels are not supported on the fused loss path")
            out.append(dict(x, labels=oh.argmax(-1) if oh.numel() else oh.new_zeros((0,), dtype=torch.long),
                            prompt_feature=x['query_feats']))
        return out

    def _batch_cf(self, video_list, is_training=True, padding_val=0.0):
        vl = [x if 'labels' in x else dict(x, labels=[0]) for x in video_list]      # NLQ batches every clip (:923)
        return super()._batch_cf(vl, is_training, padding_val)

    def _query_batch_cf(self, video_list, padding_val=0.0):
        vl = [x if 'prompt_feature' in x else dict(x, prompt_feature=x['query_feats']) for x in video_list]
        return super()._query_batch_cf(vl, padding_val)

    def prepare(self, video_list, is_training=True, gt_pad=None):
        """the host half of the step (PtTransformer.prepare) over NLQ's batch dictionaries: class indices come from the
        one-hot rows, the text from `query_feats` -- also what vilco_amd.graph.GraphedStep calls"""
        if self.training and is_training and video_list and 'labels' not in video_list[0]:
            video_list = self._with_labels(video_list)
        return super().prepare(video_list, is_training, gt_pad)

    def forward(self, video_list, task_id=-1, ensemble=False, hidden_state=False, is_training=True,
                prev_out_cls_logits=None, get_emb=False, val_qilDatasetList=None):
        if self.training and not get_emb:
            video_list = self._with_labels(video_list)
        # the reference returns losses iff self.training (:744); `is_training` only selects the padding policy and the
        # prompt mask there -- MQ's forward keys everything on is_training, so pass the module state for that decision
        return super().forward(video_list, task_id=task_id, ensemble=ensemble, hidden_state=hidden_state,
                               is_training=is_training if not self.training else True,
                               prev_out_cls_logits=prev_out_cls_logits, get_emb=get_emb,
                               val_qilDatasetList=val_qilDatasetList)
