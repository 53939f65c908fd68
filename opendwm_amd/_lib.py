"""ctypes binding of libdwm_hip.so (include/dwm_hip.h).  There is no CPU fallback:
if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DWM_HIP_LIB") or os.path.join(HERE, "libdwm_hip.so")      # DWM_HIP_LIB: another build of the same ABI (A/B measurements)
ABI_VERSION = 17

EPI_PLAIN, EPI_GEGLU, EPI_RESID, EPI_RMSHEAD = 0, 1, 2, 3
ACT_NONE, ACT_GELU_TANH, ACT_SILU, ACT_RELU = 0, 1, 2, 3

_i64, _i32, _f32, _vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class RowMap2D(C.Structure):
    _fields_ = [("rw", _i64), ("rh", _i64), ("rpitch", _i64), ("ipitch", _i64), ("origin", _i64), ("xstep", _i64)]


class GemmTnArgs(C.Structure):
    _fields_ = [
        ("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("b_rows", _i64), ("out", _vp), ("ldo", _i64),
        ("M", _i64), ("N", _i64), ("C", _i64), ("ntaps", _i32), ("split_k", _i32), ("tap_shift", _i64 * 27),
        ("workspace", _vp), ("workspace_bytes", _i64),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", _vp), ("lda", _i64), ("W", _vp), ("bias", _vp), ("C", _vp), ("ldc", _i64),
        ("M", _i64), ("N", _i64), ("K", _i64), ("epilogue", _i32), ("act", _i32),
        ("gate", _vp), ("ld_gate", _i64), ("rows_per_gate", _i64),
        ("res", _vp), ("ld_res", _i64), ("res_mod", _i64),
        ("blend", _vp), ("ld_blend", _i64), ("alpha", _vp), ("rows_per_alpha", _i64),
        ("rms_w", _vp), ("rms_ncols", _i64), ("rms_eps", _f32), ("reserved", _i32),
        ("a_map", RowMap2D), ("c_map", RowMap2D), ("ntaps", _i32), ("k_per_tap", _i32),
        ("tap_shift", _i64 * 27),
        ("workspace", _vp), ("workspace_bytes", _i64), ("split_k", _i32),
        ("C32", _vp), ("ldc32", _i64),
        ("tile", _i32),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q0", _vp), ("k0", _vp), ("v0", _vp), ("ld0", _i64),
        ("q1", _vp), ("k1", _vp), ("v1", _vp), ("ld1", _i64),
        ("o0", _vp), ("ldo0", _i64), ("o1", _vp), ("ldo1", _i64),
        ("L0", _i64), ("L1", _i64), ("n_problems", _i64),
        ("heads", _i32), ("head_dim", _i32), ("scale", _f32), ("mask_mode", _i32),
        ("pdiv", _i64 * 3), ("pmod", _i64 * 3), ("pstride", _i64 * 3),
        ("ldiv", _i64 * 2), ("lstride", _i64 * 3),
        ("mask", _vp), ("mask_G", _i64), ("group_size", _i64), ("p_per_mask", _i64),
        ("variant", _i32), ("cross", _i32), ("lse", _vp),
    ]


class AttnBwdArgs(C.Structure):
    _fields_ = [
        ("fwd", AttnArgs), ("do0", _vp), ("do1", _vp),
        ("dq0", _vp), ("dk0", _vp), ("dv0", _vp), ("ld_d0", _i64),
        ("dq1", _vp), ("dk1", _vp), ("dv1", _vp), ("ld_d1", _i64), ("delta", _vp),
    ]


class LayerNormArgs(C.Structure):
    _fields_ = [
        ("x", _vp), ("ldx", _i64), ("y", _vp), ("ldy", _i64), ("y2", _vp), ("ldy2", _i64),
        ("xsum", _vp), ("ldxsum", _i64), ("rows", _i64), ("D", _i32), ("eps", _f32),
        ("weight", _vp), ("bias", _vp),
        ("scale", _vp), ("shift", _vp), ("ld_mod", _i64), ("rows_per_mod", _i64),
        ("scale2", _vp), ("shift2", _vp),
        ("addvec", _vp), ("ld_add", _i64), ("rows_per_add", _i64),
    ]


class GnImgMap(C.Structure):
    _fields_ = [("iv", _i64), ("pn", _i64), ("s_ihi", _i64), ("s_ilo", _i64), ("s_phi", _i64)]


class GnZMap(C.Structure):
    _fields_ = [("mod", _vp), ("ld_mod", _i64), ("frames", _i32), ("videos", _i32), ("h", _i32), ("w", _i32),
                ("shift", _i32), ("zt", _i32 * 32)]


class FrameMix(C.Structure):
    _fields_ = [("n_out", _i32), ("f0", _i32 * 64), ("f1", _i32 * 64), ("w0", _f32 * 64), ("w1", _f32 * 64)]


class BlockPermuteArgs(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("block_bytes", C.c_int64), ("n", C.c_int64 * 4), ("sstride", C.c_int64 * 4)]


class RowCombineArgs(C.Structure):
    _fields_ = [
        ("a", _vp), ("lda", _i64), ("gate_a", _vp), ("ld_gate_a", _i64), ("rows_per_gate_a", _i64),
        ("coef_a", _vp), ("rows_per_coef_a", _i64), ("b", _vp), ("ldb", _i64),
        ("coef_b", _vp), ("rows_per_coef_b", _i64), ("out", _vp), ("ldo", _i64), ("rows", _i64), ("ncols", _i64),
    ]


class LayerNormBwdArgs(C.Structure):
    _fields_ = [
        ("x", _vp), ("ldx", _i64), ("addvec", _vp), ("ld_add", _i64), ("rows_per_add", _i64),
        ("dy", _vp), ("lddy", _i64), ("dy2", _vp), ("lddy2", _i64),
        ("dx", _vp), ("lddx", _i64), ("accumulate", _i32),
        ("rows", _i64), ("D", _i32), ("eps", _f32), ("weight", _vp),
        ("scale", _vp), ("scale2", _vp), ("ld_mod", _i64), ("rows_per_mod", _i64),
        ("dgamma", _vp), ("dbeta", _vp), ("dgamma2", _vp), ("dbeta2", _vp), ("ld_grad", _i64), ("grad_per_group", _i32),
    ]


# name -> (restype, argtypes); every symbol include/dwm_hip.h declares
SIGNATURES = {
    "dwm_abi_version": (_i32, []),
    "dwm_source_hash": (C.c_char_p, []),
    "dwm_gemm_bf16": (_i32, [C.POINTER(GemmArgs), _vp]),
    "dwm_gemm4w_launches": (_i64, []),
    "dwm_gemm4w_launches_general": (_i64, []),
    "dwm_attn_stream_launches": (_i64, []),
    "dwm_gemm_tn": (_i32, [C.POINTER(GemmTnArgs), _vp]),
    "dwm_attention_fwd": (_i32, [C.POINTER(AttnArgs), _vp]),
    "dwm_attention_bwd": (_i32, [C.POINTER(AttnBwdArgs), _vp]),
    "dwm_debug_tr_probe": (_i32, [_vp, _vp, _vp]),
    "dwm_layernorm": (_i32, [C.POINTER(LayerNormArgs), _vp]),
    "dwm_layernorm_x32": (_i32, [C.POINTER(LayerNormArgs), _vp]),
    "dwm_rmsnorm_heads": (_i32, [_vp, _i64, _i64, _i64, _vp, _f32, _vp]),
    "dwm_silu": (_i32, [_vp, _vp, _i64, _vp]),
    "dwm_timestep_sinusoid": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "dwm_patchify": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "dwm_unpatchify": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dwm_cfg_euler_step": (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _vp]),
    "dwm_cfg_euler_step_grouped": (_i32, [_vp, _vp, _vp, _i64, _f32, _vp, _i64, _vp]),
    "dwm_cast_f32_to_bf16": (_i32, [_vp, _vp, _i64, _vp]),
    "dwm_gemm_f32": (_i32, [C.POINTER(GemmArgs), _vp]),
    "dwm_layernorm_f32": (_i32, [C.POINTER(LayerNormArgs), _vp]),
    "dwm_groupnorm_silu_f32": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _i32, _vp, C.POINTER(RowMap2D), C.POINTER(GnImgMap), _vp]),
    "dwm_upsample2_padded_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dwm_pad_tokens_f32": (_i32, [_vp, _vp, _i64, _i32, C.POINTER(RowMap2D), _vp]),
    "dwm_softmax_rows_f32": (_i32, [_vp, _vp, _i64, _i32, _i64, _f32, _vp]),
    "dwm_unshuffle_tokens_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "dwm_avgpool2_tokens_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "dwm_cfg_multistep_f32": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp]),
    "dwm_attention_f32": (_i32, [C.POINTER(AttnArgs), _vp]),
    "dwm_silu_f32": (_i32, [_vp, _vp, _i64, _vp]),
    "dwm_timestep_sinusoid_f32": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "dwm_patchify_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "dwm_unpatchify_f32": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dwm_cfg_euler_step_f32": (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _vp, _i64, _vp]),
    "dwm_ray_features": (_i32, [_vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    "dwm_ray_features_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    "dwm_frame_affine": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "dwm_cfg_ddim_step": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _f32, _i32, _vp]),
    "dwm_cfg_multistep": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp]),
    "dwm_unshuffle_tokens": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "dwm_avgpool2_tokens": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "dwm_add_inplace": (_i32, [_vp, _vp, _i64, _vp]),
    "dwm_add_f32_inplace": (_i32, [_vp, _vp, _i64, _vp]),
    "dwm_add_f32_f32_inplace": (_i32, [_vp, _vp, _i64, _vp]),
    "dwm_groupnorm_silu": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _i32, _vp, C.POINTER(RowMap2D), _vp]),
    "dwm_groupnorm_silu_mapped": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _i32, _vp, C.POINTER(RowMap2D), C.POINTER(GnImgMap), _vp]),
    "dwm_upsample2_padded": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dwm_pad_tokens": (_i32, [_vp, _vp, _i64, _i32, C.POINTER(RowMap2D), _vp]),
    "dwm_groupnorm_stats_floats": (_i64, [_i64, _i64, _i32]),
    "dwm_groupnorm_spatial": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _i32, _vp, C.POINTER(RowMap2D),
                                     C.POINTER(GnImgMap), C.POINTER(GnZMap), _vp]),
    "dwm_frame_mix_bf16": (_i32, [_vp, _vp, _i64, C.POINTER(FrameMix), _vp]),
    "dwm_frame_mix_f32": (_i32, [_vp, _vp, _i64, C.POINTER(FrameMix), _vp]),
    "dwm_groupnorm_spatial_f32": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _i32, _vp, C.POINTER(RowMap2D),
                                         C.POINTER(GnImgMap), C.POINTER(GnZMap), _vp]),
    "dwm_softmax_rows": (_i32, [_vp, _vp, _i64, _i32, _i64, _f32, _vp]),
    # training
    "dwm_transpose_bf16": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp]),
    "dwm_segsum": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dwm_segsum_diff": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dwm_act_fwd": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "dwm_act_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "dwm_geglu_fwd": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dwm_geglu_bwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dwm_rowcombine": (_i32, [C.POINTER(RowCombineArgs), _vp]),
    "dwm_layernorm_bwd": (_i32, [C.POINTER(LayerNormBwdArgs), _vp]),
    "dwm_rmsnorm_heads_train": (_i32, [_vp, _i64, _i64, _i64, _vp, _f32, _vp, _vp]),
    "dwm_rmsnorm_heads_bwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "dwm_adamw": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp]),
    "dwm_adamw_multi": (_i32, [_vp, _vp, _vp, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp]),
    "dwm_block_permute": (_i32, [C.POINTER(BlockPermuteArgs), _vp]),
    "dwm_cast_bf16_to_f32": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp]),
    "dwm_groupnorm_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _i32, _i32, _vp, _vp, _vp,
                                 C.POINTER(RowMap2D), C.POINTER(GnImgMap), _vp]),
}

_ERR = {-1: "DWM_EINVAL (bad shape / null pointer)", -2: "DWM_EALIGN (alignment)",
        -3: "DWM_EUNSUPPORTED (shape not supported by the kernel)"}

_lib = None


def load():
    """Load (once) and return the ctypes library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64; it must be in the process before libdwm_hip.so is
    # dlopen'ed so both bind to the SAME HIP runtime (otherwise launches fail with hipErrorNoDevice).
    import torch  # noqa: F401
    if torch.cuda.is_available():
        torch.cuda.init()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m opendwm_amd.build` "
            "(there is no CPU / PyTorch fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.dwm_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libdwm_hip.so ABI {v} != binding ABI {ABI_VERSION}; rebuild")
    # a binary older than the sources next to it must not pass silently (ensure_built() never rebuilds an existing .so)
    if os.environ.get("DWM_SKIP_SOURCE_HASH") != "1":
        from .build import source_hash
        try:
            want = source_hash()
        except OSError:
            want = None                      # sources not shipped next to the binding: nothing to compare
        have = (lib.dwm_source_hash() or b"").decode()
        if want is not None and have != want:
            msg = (f"{LIB_PATH} was built from other sources (hash {have[:12]}..., tree {want[:12]}...): "
                   "run `python -m opendwm_amd.build`")
            if os.environ.get("DWM_HIP_LIB"):            # an explicitly chosen library (A/B measurements against an older / development build)
                import warnings
                warnings.warn(msg)
            else:
                raise RuntimeError(msg)
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = _ERR.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise RuntimeError(f"{what} failed: {msg}")
