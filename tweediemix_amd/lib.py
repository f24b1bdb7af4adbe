"""ctypes binding of libtmix_hip.so (the C ABI declared in include/tmix.h).

The product path has NO fallback: if the shared library is missing or a call fails, we raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtmix_hip.so")
LIB_PATH = os.environ.get("TMIX_LIB", LIB_PATH)      # dev: an A/B build from tools/build_variant.sh (still a HIP library: no fallback)

OK, EINVAL, ESHAPE, EARCH, EALIGN = 0, -1, -2, -3, -4
F32, F16, BF16 = 0, 1, 2
STEP_FUSION, STEP_PLAIN, STEP_RESAMPLE = 0, 1, 2
EPI_NONE, EPI_GEGLU, EPI_F32OUT, EPI_GELU, EPI_QUICKGELU = 0, 1, 2, 3, 4
CONV_S1, CONV_S2, CONV_UP2, CONV_T3, CONV_S2A = 0, 1, 2, 3, 4
F8_A_BLOCK_SCALES, F8_GEGLU_OUT, F8_COPY_OUT = 1, 2, 4    # tmix_gemm_desc.reserved0 flags (the first two: tmix_gemm_fp8 only)
TILE_AUTO, TILE_COUNT, TILE_COUNT_CONV = 0, 26, 7       # 6, 8..11: retired ids (run as 4, 19, 2, 1, 4)
TILE_CANDIDATES = (1, 2, 3, 4, 5, 7, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22)          # what the autotuner times by default (16, 17: phase-offset mainloop).  NOT 23: its 16x16x32 MFMAs add up a row's products in another order than the 32x32x16 tilings (which are bit-identical among themselves), so a tuner that picked it for one plan and not for its co-batched twin broke test_two_seeds_co_batched_equal_independent_runs; in situ it is level with 21 anyway.  Nor 24 (256x320 on persistent workgroups, same bits as 14): 5 % ahead hot, 5 % behind in situ
TILE_EXCLUSIVE = (13, 14, 15, 22, 24)                                  # one workgroup per CU over the whole chip: not beside a sibling chain
# gemm_conv.hip:launch substitutes these tilings when a bf16 GEMM also leaves the e4m3 copy of its output (TMIX_F8_COPY_OUT is
# compiled into the tilings with registers to spare).  The plan builder applies the same map to the DESCRIPTOR, so the number of
# row-statistics partials the consumers are told (tmix_gemm_stats_parts) is that of the kernel that really runs.
F8COPY_TILE_ALT = {6: 4, 8: 7, 9: 2, 10: 1, 11: 4, 14: 12, 19: 12, 20: 12, 21: 12, 22: 12, 23: 12, 24: 12, 25: 12, 26: 12}
TILE_CONV_HALO = 26                                              # gemm_convh.hip: stride-1 convolutions with the halo patch in LDS (conv launches only)
TILE_LW = (19, 20, 21)                                           # loader-wave tilings: GEMM only, except 20 (a conv launch runs 19 / 21 as tiling 12)

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    """mirror of tmix_gemm_desc"""
    _fields_ = [("A", vp), ("lda", i64), ("strideA", i64),
                ("W", vp), ("ldw", i64), ("strideW", i64),
                ("C", vp), ("ldc", i64), ("strideC", i64),
                ("bias", vp), ("strideBias", i64),
                ("residual", vp), ("ldr", i64), ("strideR", i64),
                ("rowgroup_bias", vp), ("rows_per_group", i32),
                ("Ct", vp), ("ldct", i64), ("strideCt", i64),
                ("n_trans_begin", i32),
                ("M", i32), ("N", i32), ("K", i32), ("batch", i32),
                ("epilogue", i32), ("tile_cfg", i32),
                ("row_stats_out", vp), ("strideStatsOut", i64), ("ldStatsOut", i64),
                ("ln_stats", vp), ("strideLnStats", i64), ("ldLnStats", i64),
                ("ln_colsum", vp), ("strideLnColsum", i64), ("ln_inv_c", f32), ("ln_eps", f32),
                ("ln_parts", C.c_int32), ("reserved0", C.c_int32),
                ("col_stats_out", vp), ("w_period", C.c_int32), ("reserved1", C.c_int32)]


class ConvDesc(C.Structure):
    """mirror of tmix_conv_desc"""
    _fields_ = [("X", vp), ("Wt", vp), ("Y", vp), ("bias", vp), ("batch_bias", vp), ("residual", vp),
                ("B", i32), ("H", i32), ("W", i32), ("Cin", i32), ("Cout", i32), ("mode", i32), ("tile_cfg", i32),
                ("batch_bias_images", i32), ("col_stats_out", vp), ("S1", vp), ("S2", vp), ("S1_channels", i32), ("S2_channels", i32)]


SIGNATURES = {
    "tmix_version": (C.c_int, []),
    "tmix_last_error_string": (C.c_char_p, []),
    "tmix_check_device": (C.c_int, []),
    "tmix_env_refresh": (None, []),
    "tmix_prof_begin": (C.c_int, [vp, C.c_int, C.c_int]),
    "tmix_prof_end": (C.c_int, []),
    "tmix_fused_tweedie_step": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, i64, C.c_int,
                                          f32, f32, f32, f32, f32, C.c_int, vp]),
    "tmix_fused_tweedie_step_dev": (C.c_int, [vp, vp, C.c_int, vp, i64, vp, vp, C.c_int, C.c_int, i64, C.c_int, C.c_int,
                                              C.c_int, vp, vp]),
    "tmix_step_prologue": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, i64, vp]),
    "tmix_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "tmix_gemm_prefetch_next": (C.c_int, [vp, i64, vp]),
    "tmix_gemm_fp8": (C.c_int, [C.POINTER(GemmDesc), vp, vp, vp]),
    "tmix_gemm_q_cross_attn": (C.c_int, [C.POINTER(GemmDesc), vp, i64, i64, vp, i64, i64, vp, i64, C.c_int, C.c_int, f32, vp]),
    "tmix_quantize_fp8_rows": (C.c_int, [vp, i64, vp, i64, vp, i64, C.c_int, vp]),
    "tmix_conv3x3_nhwc": (C.c_int, [C.POINTER(ConvDesc), vp]),
    "tmix_conv3x3_nhwc_fp8": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp]),
    "tmix_conv_in": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tmix_conv_in_pre": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "tmix_softmax_rows": (C.c_int, [vp, i64, vp, i64, i64, C.c_int, f32, vp]),
    "tmix_softmax_rows_causal": (C.c_int, [vp, i64, vp, i64, i64, C.c_int, f32, C.c_int, vp]),
    "tmix_softmax_rows_masked": (C.c_int, [vp, i64, vp, i64, i64, C.c_int, C.c_int, f32, vp]),
    "tmix_affine_clamp": (C.c_int, [vp, vp, i64, f32, f32, f32, f32, vp]),
    "tmix_conv_out": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tmix_attn_fwd": (C.c_int, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64,
                                C.c_int, C.c_int, C.c_int, C.c_int, f32, vp]),
    "tmix_attn_fwd_f8": (C.c_int, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, vp, i64,
                                   C.c_int, C.c_int, C.c_int, C.c_int, f32, vp]),
    "tmix_attn_split_ws_bytes": (i64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "tmix_attn_fwd_ws": (C.c_int, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64,
                                   C.c_int, C.c_int, C.c_int, C.c_int, f32, vp, i64, vp]),
    "tmix_attn_fwd_f8_ws": (C.c_int, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, vp, i64,
                                      C.c_int, C.c_int, C.c_int, C.c_int, f32, vp, i64, vp]),
    "tmix_groupnorm_nhwc_launches": (C.c_int, [i64, C.c_int, C.c_int]),
    "tmix_groupnorm_ws_chunks": (C.c_int, [i64]),
    "tmix_groupnorm_ws_floats": (i64, [C.c_int, C.c_int, C.c_int]),
    "tmix_groupnorm_nhwc": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, i64, C.c_int, f32,
                                      C.c_int, vp]),
    "tmix_groupnorm_nhwc_pre": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, i64, C.c_int, f32,
                                          C.c_int, vp, C.c_int, vp, C.c_int, vp]),
    "tmix_groupnorm_nhwc_pre_f8": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, f32,
                                             C.c_int, vp, C.c_int, vp, C.c_int, vp]),
    "tmix_layernorm": (C.c_int, [vp, vp, vp, vp, i64, C.c_int, f32, vp]),
    "tmix_zero": (C.c_int, [vp, i64, vp]),
    "tmix_temporal_attn": (C.c_int, [vp, i64, vp, i64, C.c_int, C.c_int, i64, C.c_int, f32, vp]),
    "tmix_vpred_step": (C.c_int, [vp, vp, vp, C.c_int, i64, f32, f32, f32, f32, f32, vp]),
    "tmix_frame_inject": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i64, C.c_int, f32, f32, vp]),
    "tmix_gemm_tile_shape": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tmix_gemm_stats_parts": (C.c_int, [C.c_int, C.c_int]),
    "tmix_concat_channels": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, i64, vp]),
    "tmix_lora_down": (C.c_int, [vp, i64, C.c_int, i64, vp, C.c_int, C.c_int, vp, vp, f32, vp, i64, vp]),
    "tmix_timestep_embedding": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "tmix_linear_small": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tmix_linear_small_sections": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "tmix_conv3x3_f32": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 7 + [vp]),
    "tmix_adaptive_avgpool_f32": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tmix_linear_f32": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tmix_i2v_temporal_encoder": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64] + [vp] * 11 + [vp]),
}

_lib = None


class TmixError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libtmix_hip.so and attach prototypes.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: libtmix_hip.so needs libamdhip64.so.7 and has to bind to the SAME
    # HIP runtime instance torch uses (streams and device pointers are shared); torch ships its own copy.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
                          f"or make -C tweediemix_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI and the header diverge
        fn.restype = res
        fn.argtypes = args
    if lib.tmix_version() != 100:
        raise ImportError(f"libtmix_hip.so ABI version {lib.tmix_version()} != 100")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().tmix_last_error_string().decode("utf-8", "replace")
        raise TmixError(f"{what} failed with code {rc}: {msg}")
