"""ctypes binding of libvilco_hip.so (C ABI: include/vilco_hip.h).

The product path has NO fallback: if the library is missing or a kernel returns a status, we
raise.  `load()` is lazy so that CPU-only tooling (config parsing, state_dict surgery, the
`-m "not gpu"` symbol test) can import the package without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VILCO_HIP_LIB") or os.path.join(_HERE, "libvilco_hip.so")   # override: tools/lab builds

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
TAP_NONE, TAP_A, TAP_B = 0, 1, 2

c_fp = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
i32, i64, f32, sz = C.c_int32, C.c_int64, C.c_float, C.c_size_t


class GemmDesc(C.Structure):
    """Mirror of `vilco_gemm_desc` (include/vilco_hip.h)."""
    _fields_ = [
        ("A", c_fp), ("B", c_fp), ("C", c_fp),
        ("M", i32), ("N", i32), ("K", i32),
        ("a_kcontig", i32), ("b_kcontig", i32),
        ("lda", i64), ("ldb", i64), ("ldc", i64),
        ("batch_outer", i32), ("batch_inner", i32),
        ("sAo", i64), ("sAi", i64), ("sBo", i64), ("sBi", i64), ("sCo", i64), ("sCi", i64),
        ("tap_operand", i32), ("tapC", i32), ("tapT", i32),
        ("precision", i32),
        ("alpha", f32), ("beta", f32),
        ("bias", c_fp), ("preact", c_fp), ("act", i32),
        ("row_len", c_fp), ("rowT", i32),
        ("colscale", c_fp), ("residual", c_fp), ("res_masked", i32),
        ("workspace", c_fp), ("workspace_bytes", sz),
        ("a_planes", c_fp), ("b_planes", c_fp),
        ("band", i32), ("bandT", i32),
        ("drop_p", f32), ("drop_seed", C.c_uint32),
        ("amax_out", c_fp),
        ("a_amax", c_fp), ("a_namax", i32), ("b_amax", c_fp), ("b_namax", i32),
        ("a_planes_seq", i32), ("b_planes_seq", i32),
        ("row_mask", c_fp),
    ]


class LossDesc(C.Structure):
    """Mirror of `vilco_loss_desc`."""
    _fields_ = [("logits", c_fp), ("offsets", c_fp), ("level_scale", c_fp), ("points", c_fp), ("row_level", c_fp),
                ("row_pos", c_fp), ("level_len", c_fp), ("gt", c_fp), ("gauss", c_fp), ("loss_norm", c_fp),
                ("B", i32), ("R", i32), ("C", i32), ("L", i32), ("Nmax", i32),
                ("center_radius", f32), ("label_smoothing", f32), ("momentum", f32), ("loss_weight", f32),
                ("al_weight", f32), ("use_al", i32)]


class AttnAmaxIn(C.Structure):
    """Mirror of `vilco_attn_amax_in`."""
    _fields_ = [("q", c_fp), ("nq", i32), ("k", c_fp), ("nk", i32), ("v", c_fp), ("nv", i32), ("dout", c_fp), ("ndo", i32)]


class PackItem(C.Structure):
    """Mirror of `vilco_pack_item`."""
    _fields_ = [("src", c_fp), ("rows", i64), ("cols", i64), ("ld", i64), ("planes", c_fp), ("planes_bytes", sz),
                ("nbatch", i32), ("batch_stride", i64), ("relshift", i32), ("amax", c_fp), ("namax", i32), ("seq_len", i32)]


# name -> (restype, argtypes); must list every symbol include/vilco_hip.h declares
SIGNATURES = {
    "vilco_status_str": (C.c_char_p, [C.c_int]),
    "vilco_version": (C.c_char_p, []),
    "vilco_sync_timeouts_read": (C.c_int, []),
    "vilco_defer_set": (C.c_int, [i32]),
    "vilco_defer_pending": (i64, []),
    "vilco_defer_flush": (C.c_int, [c_fp]),
    "vilco_gemm_workspace": (sz, [C.POINTER(GemmDesc)]),
    "vilco_gemm": (C.c_int, [C.POINTER(GemmDesc), c_fp]),
    "vilco_gemm_group": (C.c_int, [C.POINTER(GemmDesc), i32, c_fp]),
    "vilco_gemm_amax_parts": (i32, [C.POINTER(GemmDesc)]),
    "vilco_gemm_force": (C.c_int, [i32, i32]),
    "vilco_gemm_set_fixup": (C.c_int, [i32]),
    "vilco_gemm_set_gl": (C.c_int, [i32]),
    "vilco_gemm_set_tail128": (C.c_int, [i32]),
    "vilco_gemm_set_skinny": (C.c_int, [i32]),
    "vilco_gemm_config_gen": (i64, []),
    "vilco_gemm_profile_begin": (C.c_int, []),
    "vilco_gemm_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "vilco_gemm_profile_records": (C.c_int64, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int64]),
    "vilco_pack_bytes": (sz, [i64, i64, i32]),
    "vilco_pack_item_bytes": (sz, [C.POINTER(PackItem), i32]),
    "vilco_pack": (C.c_int, [c_fp, i64, i64, i64, i32, c_fp, sz, c_fp]),
    "vilco_pack_many": (C.c_int, [C.POINTER(PackItem), i32, i32, c_fp]),
    "vilco_layernorm_fwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i64, i32, f32, i32, c_fp]),
    "vilco_layernorm_fwd_amax": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i64, i32, f32, i32, c_fp, C.POINTER(i32), c_fp]),
    "vilco_layernorm_planes_bytes": (sz, [i64, i32, i32]),
    "vilco_layernorm_fwd_planes": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i64, i32, f32, i32, c_fp, C.POINTER(i32), c_fp, sz,
                                             i32, c_fp, i64, c_fp]),
    "vilco_layernorm_bwd_workspace": (sz, [i64, i32]),
    "vilco_layernorm_bwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i64, i32,
                                      i32, c_fp, sz, c_fp]),
    "vilco_layernorm_bwd_res": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i64, i32,
                                          i32, c_fp, sz, c_fp]),
    "vilco_layernorm_bwd_res_amax": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i64, i32,
                                               i32, c_fp, sz, c_fp, C.POINTER(i32), c_fp]),
    "vilco_dwconv3_fwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp]),
    "vilco_dwconv3_bwd_workspace": (sz, [i32, i32, i32, i32]),
    "vilco_dwconv3_bwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, sz, c_fp]),
    "vilco_maxpool3s2_fwd": (C.c_int, [c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "vilco_maxpool3s2_bwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "vilco_softmax_fwd": (C.c_int, [c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "vilco_softmax_bwd": (C.c_int, [c_fp, c_fp, i32, i32, i32, i32, c_fp]),
    "vilco_relshift_add": (C.c_int, [c_fp, c_fp, f32, i32, i32, i32, c_fp]),
    "vilco_relshift_bwd": (C.c_int, [c_fp, c_fp, f32, i32, i32, i32, c_fp]),
    "vilco_attn_supported": (C.c_int, [i32]),
    "vilco_attn_fwd_workspace": (sz, [i32, i32, i32, i32, i32, i32]),
    "vilco_attn_amax_parts": (i32, [i32, i32, i32, i32, i32, i32, i32, f32, i32]),
    "vilco_attn_fwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, f32, i32, i32, i32, f32,
                                 C.c_uint32, C.POINTER(AttnAmaxIn), c_fp, c_fp, sz, c_fp]),
    "vilco_attn_fwd_planes": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, f32, i32, i32, i32, f32,
                                        C.c_uint32, C.POINTER(AttnAmaxIn), c_fp, c_fp, sz, c_fp, sz, c_fp]),
    "vilco_attn_planes_supported": (i32, [i32, i32, i32, i32, i32, i32, f32]),
    "vilco_attn_bwd_workspace": (sz, [i32, i32, i32, i32, i32, i32]),
    "vilco_attn_bwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32,
                                 i32, i32, i32, f32, i32, i32, i32, f32, C.c_uint32, C.POINTER(AttnAmaxIn), c_fp, c_fp, c_fp, c_fp, c_fp, sz,
                                 c_fp]),
    "vilco_attn_dsplanes_bytes": (sz, [i32, i32, i32]),
    "vilco_xl_scores_workspace": (sz, [i32, i32, i32, i32]),
    "vilco_xl_scores": (C.c_int, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp, sz, c_fp]),
    "vilco_attn_bwd_dsplanes": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32,
                                          i32, i32, i32, f32, i32, i32, i32, f32, C.c_uint32, C.POINTER(AttnAmaxIn), c_fp, c_fp, c_fp, c_fp,
                                          c_fp, sz, c_fp, sz, c_fp]),
    "vilco_decode_workspace": (sz, [i32, i32]),
    "vilco_decode": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, f32, f32, c_fp, c_fp, c_fp, c_fp, c_fp, sz, c_fp]),
    "vilco_scale_add_fwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp]),
    "vilco_colsum_workspace": (sz, [i64, i32]),
    "vilco_scale_add_bwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, c_fp, c_fp, i32, i32, i32,
                                      c_fp, sz, c_fp]),
    "vilco_scale_add_bwd_amax": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, c_fp, c_fp, i32, i32, i32,
                                           c_fp, sz, c_fp, C.POINTER(i32), c_fp]),
    "vilco_dropout": (C.c_int, [c_fp, c_fp, i64, f32, C.c_uint32, C.c_uint64, c_fp]),
    "vilco_attn_dropout_mask": (C.c_int, [c_fp, i64, i32, f32, C.c_uint32, c_fp]),
    "vilco_seed_word_set": (C.c_int, [C.c_uint32, c_fp]),
    "vilco_seed_word_bump": (C.c_int, [c_fp]),
    "vilco_seed_word_get": (C.c_int, [C.POINTER(C.c_uint32)]),
    "vilco_axpby": (C.c_int, [c_fp, c_fp, c_fp, f32, f32, i64, c_fp]),
    "vilco_act_bwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i64, i32, f32, C.c_uint32, c_fp, sz, c_fp]),
    "vilco_act_bwd_amax": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i64, i32, f32, C.c_uint32, c_fp, sz, c_fp,
                                     C.POINTER(i32), c_fp]),
    "vilco_act_bwd_planes": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i64, i32, f32, C.c_uint32, c_fp, sz, c_fp,
                                       C.POINTER(i32), c_fp, i32, c_fp, sz, c_fp, c_fp]),
    "vilco_act_bwd_planes_bytes": (sz, [i64, i32, i32]),
    "vilco_act_bwd_planes_seq": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i64, i32, f32, C.c_uint32, c_fp, sz, c_fp,
                                           C.POINTER(i32), c_fp, i32, c_fp, sz, i32, c_fp, c_fp]),
    "vilco_colsum": (C.c_int, [c_fp, c_fp, i64, i32, c_fp, sz, c_fp]),
    "vilco_mask_rows": (C.c_int, [c_fp, c_fp, i32, i32, i32, c_fp]),
    "vilco_add_pe": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "vilco_transpose2d": (C.c_int, [c_fp, c_fp, i32, i32, i32, c_fp]),
    "vilco_permute3": (C.c_int, [c_fp, c_fp, i32, i32, i32, i64, i64, i64, i64, c_fp]),
    "vilco_grad_norm": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, f32, c_fp, c_fp, c_fp]),
    "vilco_optim_step": (C.c_int, [i32, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), i32,
                                   f32, f32, f32, f32, c_fp, c_fp, c_fp]),
    "vilco_optim_step_amax": (C.c_int, [i32, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), i32,
                                        f32, f32, f32, f32, c_fp, c_fp, c_fp, c_fp]),
    "vilco_optim_step_dev": (C.c_int, [i32, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), i32,
                                       f32, f32, f32, f32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "vilco_store_f32": (C.c_int, [c_fp, C.POINTER(f32), i32, c_fp]),
    "vilco_qkv_pre_supported": (C.c_int, [i32]),
    "vilco_qkv_pre_amax_parts": (C.c_int, [i32, i32, i32]),
    "vilco_qkv_pre_fwd": (C.c_int, [c_fp, c_fp, c_fp, C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_fp), c_fp, c_fp,
                                    C.POINTER(c_fp), c_fp, c_fp, C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_fp), i32, i32,
                                    i32, i32, f32, f32, c_fp]),
    "vilco_qkv_pre_bwd_workspace": (sz, [i32, i32, i32, i32]),
    "vilco_qkv_pre_bwd": (C.c_int, [c_fp, C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_fp),
                                    C.POINTER(c_fp), c_fp, c_fp, C.POINTER(c_fp), c_fp, c_fp, i32, i32, i32, i32, c_fp, sz,
                                    c_fp]),
    "vilco_mq_loss_workspace": (sz, [i32, i32, i32]),
    "vilco_mq_loss_fwd": (C.c_int, [C.POINTER(LossDesc), c_fp, c_fp, c_fp, c_fp, sz, c_fp]),
    "vilco_mq_loss_bwd": (C.c_int, [C.POINTER(LossDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                    c_fp]),
    "vilco_cl_penalty": (C.c_int, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, f32, i32, c_fp, c_fp, c_fp]),
    "vilco_nms_workspace": (sz, [i64, i32]),
    "vilco_nms_1d": (C.c_int, [c_fp, c_fp, c_fp, i32, i64, f32, c_fp, c_fp, c_fp, sz, c_fp]),
    "vilco_softnms_1d": (C.c_int, [c_fp, c_fp, c_fp, i32, i64, f32, f32, f32, i32, i64, c_fp, c_fp, c_fp,
                                   c_fp, sz, c_fp]),
    "vilco_nms_set_kernel": (C.c_int, [i32]),
    "vilco_nms_last_kernels": (C.c_int, []),
}

_lib = None


def load():
    """Load libvilco_hip.so (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvilco_hip.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C vilco_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError(load().vilco_status_str(int(rc)).decode())


def version():
    return load().vilco_version().decode()
