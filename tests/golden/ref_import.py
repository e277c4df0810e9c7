"""TEST INFRASTRUCTURE ONLY: import the *reference* MQ model from /root/reference in THIS
container, to pin the oracle and to generate golden vectors (SURVEY.md section 8c recipe).

Nothing from the reference is copied: a scratch directory of symlinks gives the reference
the CWD layout it expects (`libs/`, `configs/xlnet_config_<D>.json` resolved relative to
CWD, backbones.py:132), plus extra tiny XLNet JSON configs that we author for golden sizes.
This module cannot run on the GPU box (no /root/reference there) and is never imported by
the product path.
"""
import json
import os
import sys

REF_ROOT = "/root/reference/MQ"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SCRATCH = "/tmp/vilco_ref_scratch/MQ"


def xlnet_json(d_model, n_head, d_inner=None, dropout=0.0):
    """An XLNet config dict in the shape of MQ/configs/xlnet_config_*.json."""
    return {
        "architectures": ["XLNetLMHeadModel"], "attn_type": "bi", "bi_data": False,
        "clamp_len": -1, "d_head": d_model // n_head, "d_inner": d_inner or 2 * d_model,
        "d_model": d_model, "dropout": dropout, "end_n_top": 5, "ff_activation": "gelu",
        "initializer_range": 0.02, "layer_norm_eps": 1e-12, "mem_len": 256,
        "model_type": "xlnet", "n_head": n_head, "n_layer": 1, "reuse_len": None,
        "same_length": False, "start_n_top": 5, "summary_activation": "tanh",
        "summary_last_dropout": 0.1, "summary_type": "last", "summary_use_proj": True,
        "untie_r": True, "vocab_size": 32,
    }


def available():
    return os.path.isdir(REF_ROOT)


def setup(extra_xlnet=((64, 4), (128, 4), (2304, 16))):
    """Create the scratch tree, chdir into it, install shims, import the reference.

    Returns the imported `libs` package (libs.modeling, libs.utils, libs.core)."""
    if not available():
        raise RuntimeError("reference tree not present")
    os.makedirs(os.path.join(SCRATCH, "configs"), exist_ok=True)
    link = os.path.join(SCRATCH, "libs")
    if not os.path.islink(link):
        os.symlink(os.path.join(REF_ROOT, "libs"), link)
    for f in os.listdir(os.path.join(REF_ROOT, "configs")):
        dst = os.path.join(SCRATCH, "configs", f)
        if not os.path.lexists(dst):
            os.symlink(os.path.join(REF_ROOT, "configs", f), dst)
    for d_model, n_head in extra_xlnet:
        dst = os.path.join(SCRATCH, "configs", "xlnet_config_%d.json" % d_model)
        if not os.path.lexists(dst):
            with open(dst, "w") as f:
                json.dump(xlnet_json(d_model, n_head), f)
    os.chdir(SCRATCH)

    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))   # nms_1d_cpu.so (reference build)
    sys.path.insert(0, SCRATCH)
    sys.path.insert(0, REPO)
    from oracle import build_ref
    assert build_ref.build() is not None, "reference nms extension failed to build"

    # names the transformers-4.27-era XLNet fork imports from transformers.modeling_utils
    import transformers.modeling_utils as mu
    from transformers import pytorch_utils
    from transformers.models.xlnet import modeling_xlnet as hx
    if not hasattr(mu, "apply_chunking_to_forward"):
        mu.apply_chunking_to_forward = pytorch_utils.apply_chunking_to_forward
    for new, old in (("PoolerStartLogits", "XLNetPoolerStartLogits"),
                     ("PoolerEndLogits", "XLNetPoolerEndLogits"),
                     ("PoolerAnswerClass", "XLNetPoolerAnswerClass"),
                     ("SequenceSummary", "XLNetSequenceSummary")):
        if not hasattr(mu, new):
            setattr(mu, new, getattr(hx, old))

    import libs.utils      # noqa: F401  (import order matters: circular import)
    import libs.modeling   # noqa: F401
    import libs.core.config  # noqa: F401
    import libs
    return libs


def make_cfg(libs, **over):
    """DEFAULTS merged with overrides, the way load_config does (core/config.py:177-204)."""
    import copy
    cfgmod = libs.core.config
    cfg = copy.deepcopy(over)
    cfgmod._merge(copy.deepcopy(cfgmod.DEFAULTS), cfg)
    return cfgmod._update_config(cfg)
