"""Config surface of the MQ driver, accepted unchanged (reference: MQ/libs/core/config.py:4-204).

`DEFAULTS` is interface data: every key/value a reference YAML may rely on.  `load_config` merges a
YAML over it (missing keys are filled recursively, present keys win) and copies the dataset dims and
the train/test/cl sections into cfg['model'] -- the kwargs of `make_meta_arch` (train_cl.py:135).
The table below was dumped from the imported reference (tests/golden/ref_import.py) and is pinned
by tests/test_config.py against the golden copy.
"""
import copy

import yaml

DEFAULTS = {'init_rand_seed': 765421321,
 'dataset_name': 'epic',
 'devices': ['cuda:0'],
 'train_split': ('training',),
 'val_split': ('validation',),
 'model_name': 'LocPointTransformer',
 'dataset': {'feat_stride': 16,
             'num_frames': 32,
             'default_fps': None,
             'input_dim': 2304,
             'num_classes': 97,
             'downsample_rate': 1,
             'max_seq_len': 2304,
             'trunc_thresh': 0.5,
             'crop_ratio': None,
             'force_upsampling': False,
             'use_narration': False,
             'narration_feat_folder': None,
             'use_text': False,
             'text_feat_folder': None,
             'max_text_len': 128,
             'output_format': 'concat'},
 'loader': {'batch_size': 8, 'num_workers': 2},
 'model': {'use_xl': True,
           'backbone_type': 'convTransformer',
           'fpn_type': 'identity',
           'backbone_arch': (2, 2, 5),
           'scale_factor': 2,
           'regression_range': [(0, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 10000)],
           'n_head': 4,
           'n_mha_win_size': -1,
           'embd_kernel_size': 3,
           'embd_dim': 512,
           'embd_with_ln': True,
           'fpn_dim': 512,
           'fpn_with_ln': True,
           'fpn_start_level': 0,
           'head_dim': 512,
           'head_kernel_size': 3,
           'head_num_layers': 3,
           'head_with_ln': True,
           'max_buffer_len_factor': 6.0,
           'use_abs_pe': False,
           'use_rel_pe': False,
           'use_cross_modal': False,
           'n_txt_in': 768},
 'train_cfg': {'center_sample': 'radius',
               'center_sample_radius': 1.5,
               'loss_weight': 1.0,
               'cls_prior_prob': 0.01,
               'init_loss_norm': 2000,
               'clip_grad_l2norm': -1,
               'head_empty_cls': [],
               'dropout': 0.0,
               'droppath': 0.1,
               'label_smoothing': 0.0,
               't_c_alpha': 0.8,
               'use_dcn': False,
               'dcn_start_layer': -1,
               'use_us_fpn': False,
               'al_loss_weight': 0.0,
               'cont_loss_weight': 0.0,
               'seg_loss_weight': 0.0,
               'imp_loss_weight': 0.0,
               'temperature': 0.07,
               'queue_size': 256,
               'length_theta': 0.2,
               'use_trident_head': False,
               'num_bins': 16,
               'iou_weight_power': 1.0},
 'test_cfg': {'pre_nms_thresh': 0.001,
              'pre_nms_topk': 5000,
              'iou_threshold': 0.1,
              'min_score': 0.01,
              'max_seg_num': 1000,
              'nms_method': 'soft',
              'nms_sigma': 0.5,
              'duration_thresh': 0.05,
              'multiclass_nms': True,
              'ext_score_file': None,
              'voting_thresh': 0.75},
 'cl_cfg': {'name': None,
            'memory_size': 0,
            'pkl_file': './data/ego4d/ego4d_mq_query_incremental_22_all.pkl',
            'random_order': False,
            'reg_lambda': 0,
            'type_sampling': 'icarl',
            'path_memory': 'path_memory.pkl',
            'adv_lambda': 0,
            'prompt_pool': False,
            'pool_size': 0,
            'topk': 4,
            'length': 20,
            'embed_dim': 768,
            'narration_ssl': False,
            'narration_dim': 512,
            'ssl_factor': 0.01,
            'use_adapt': False,
            'adapt_blocks': []},
 'opt': {'type': 'AdamW',
         'momentum': 0.9,
         'weight_decay': 0.0,
         'learning_rate': 0.001,
         'epochs': 30,
         'warmup': True,
         'warmup_epochs': 5,
         'schedule_type': 'cosine',
         'schedule_steps': [],
         'schedule_gamma': 0.1}}


def _merge(src, dst):
    """fill `dst` with whatever `src` has and `dst` lacks (recursively for dict values)"""
    for key, val in src.items():
        if key not in dst:
            dst[key] = val
        elif isinstance(val, dict):
            _merge(val, dst[key])


def load_default_config():
    return DEFAULTS


def _update_config(config):
    model = config["model"]
    for k in ("input_dim", "num_classes", "max_seq_len"):
        model[k] = config["dataset"][k]
    for k in ("train_cfg", "test_cfg", "cl_cfg"):
        model[k] = config[k]
    return config


def load_config(config_file, defaults=DEFAULTS):
    with open(config_file, "r") as fd:
        config = yaml.load(fd, Loader=yaml.FullLoader)
    _merge(defaults, config)
    return _update_config(config)


def make_config(**overrides):
    """DEFAULTS (deep-copied) under a dict of overrides -- `load_config` without the YAML file."""
    cfg = copy.deepcopy(overrides)
    _merge(copy.deepcopy(DEFAULTS), cfg)
    return _update_config(cfg)
