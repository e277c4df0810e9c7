"""CPU: the C-ABI library loads and exports every symbol include/vilco_hip.h declares; argument
validation paths that need no GPU; the config surface equals the reference DEFAULTS."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vilco_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vilco_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from vilco_amd import _lib
    names = declared_symbols()
    assert len(names) >= 25
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "missing export " + n


def test_status_strings_and_version():
    from vilco_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.vilco_version()
    assert lib.vilco_status_str(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert lib.vilco_status_str(code).startswith(b"vilco_hip")


def test_bad_arguments_are_rejected_without_a_gpu():
    from vilco_amd import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.vilco_gemm(ctypes.byref(d), None) == -1              # null operands
    assert lib.vilco_layernorm_fwd(None, None, None, None, None, None, 4, 64, 1e-5, 0, None) == -1
    assert lib.vilco_nms_1d(None, None, None, -1, 0, 0.5, None, None, None, 0, None) == -1
    with pytest.raises(RuntimeError, match="bad argument"):
        _lib.check(-1)
    # an operand matrix of 2^31 bytes or more is refused (the kernel's staging loads carry 32-bit byte offsets); the check
    # comes before anything touches memory, so dummy addresses do
    d = _lib.GemmDesc()
    d.A = d.B = d.C = 4096
    d.M, d.N, d.K = 70000, 128, 20000
    d.a_kcontig = d.b_kcontig = 1
    d.lda = d.ldb = 20000
    d.ldc = 128
    d.batch_outer = d.batch_inner = 1
    d.precision = 3
    assert lib.vilco_gemm(ctypes.byref(d), None) == -2
    d.M = 7000                                                  # the same call with a legal size gets as far as the workspace check
    assert lib.vilco_gemm(ctypes.byref(d), None) == -4


def test_ops_refuse_cpu_tensors():
    import torch
    from vilco_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))


def test_config_defaults_match_reference():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import ref_import
    from vilco_amd.core import config as c
    cfg = c.make_config(dataset=dict(input_dim=96))
    assert cfg["model"]["input_dim"] == 96 and cfg["model"]["train_cfg"] is cfg["train_cfg"]
    if not ref_import.available():
        pytest.skip("reference not present")
    libs = ref_import.setup()
    assert c.DEFAULTS == libs.core.config.DEFAULTS


def test_state_dict_is_drop_in():
    """same keys and shapes as the golden reference state_dict; loads strictly"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_util import golden_cfg, load_golden, xlnet_json
    import vilco_amd.modeling as vm
    for name in ("xl", "prompt"):
        gold = load_golden(name)
        m = golden_cfg(gold)
        kw = dict(m, xlnet_config=xlnet_json(m['embd_dim'], 4)) if m['use_xl'] else m
        model = vm.make_meta_arch('LocPointTransformer', **kw)
        assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == \
               {k: tuple(v.shape) for k, v in gold['state_dict'].items()}
        model.load_state_dict(gold['state_dict'], strict=True)


def test_producer_plane_entry_points_validate_on_the_host():
    """the producer-written-planes entry points (round 4): buffer sizes equal the pack's for the same tensor, and bad layouts are
    refused before anything is launched"""
    from vilco_amd import _lib
    lib = _lib.load()
    for rows, C in ((4608, 1024), (100, 96), (77, 32)):
        assert lib.vilco_layernorm_planes_bytes(rows, C, 0) == lib.vilco_pack_bytes(rows, C, 3)
    it = _lib.PackItem()
    it.rows, it.cols, it.ld, it.nbatch, it.seq_len = 2 * 4541, 1024, 1024, 1, 4541
    assert lib.vilco_layernorm_planes_bytes(2 * 4541, 1024, 4541) == lib.vilco_pack_item_bytes(ctypes.byref(it), 3)
    x = 4096                                            # dummy, suitably aligned addresses: every check below precedes the launch
    # natural planes need C % 32 == 0, the convs' image whole sequences; a short buffer is a workspace error
    assert lib.vilco_layernorm_fwd_planes(x, None, None, x, x, x, 64, 40, 1e-5, 0, None, None, x, 1 << 30, 0, None, 0, None) == -2
    assert lib.vilco_layernorm_fwd_planes(x, None, None, x, x, x, 65, 64, 1e-5, 0, None, None, x, 1 << 30, 16, None, 0, None) == -2
    assert lib.vilco_layernorm_fwd_planes(x, None, None, x, x, x, 64, 64, 1e-5, 0, None, None, x, 128, 0, None, 0, None) == -4
    # activation backward: planes need the partial maxima of dy, C % 32 == 0 and a buffer of vilco_pack_bytes
    assert lib.vilco_act_bwd_planes(x, None, None, None, 0, None, 0, 64, 64, 0.0, 0, None, 0, None, None, None, 0, x, 1 << 30, None, None) == -1
    assert lib.vilco_act_bwd_planes(x, None, None, None, 0, None, 0, 64, 40, 0.0, 0, None, 0, None, None, x, 4, x, 1 << 30, None, None) == -1
    assert lib.vilco_act_bwd_planes(x, None, None, None, 0, None, 0, 64, 64, 0.0, 0, None, 0, None, None, x, 4, x, 128, None, None) == -4
    # a row mask on a batched product is not supported
    d = _lib.GemmDesc()
    d.A = d.B = d.C = d.row_mask = 4096
    d.M, d.N, d.K = 128, 128, 64
    d.a_kcontig = d.b_kcontig = 1
    d.lda = d.ldb = 64
    d.ldc = 128
    d.batch_outer, d.batch_inner, d.precision = 2, 1, 3
    assert lib.vilco_gemm(ctypes.byref(d), None) == -2
