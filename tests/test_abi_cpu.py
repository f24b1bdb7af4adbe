"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/tmix.h declares;
the ctypes mirrors of the descriptor structs have the C layout."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "tmix.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tmix_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tweediemix_amd import lib
    l = lib.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(l, s), f"{s} declared in include/tmix.h but not exported"
        assert s in lib.SIGNATURES, f"{s} has no ctypes prototype"
    assert l.tmix_version() == 100
    assert isinstance(l.tmix_last_error_string(), bytes)


def test_descriptor_layouts():
    from tweediemix_amd import lib
    # tmix_gemm_desc: 8-byte aligned fields in declaration order (see include/tmix.h)
    assert lib.GemmDesc.A.offset == 0 and lib.GemmDesc.W.offset == 24 and lib.GemmDesc.C.offset == 48
    assert lib.GemmDesc.bias.offset == 72 and lib.GemmDesc.residual.offset == 88
    assert lib.GemmDesc.rowgroup_bias.offset == 112 and lib.GemmDesc.rows_per_group.offset == 120
    assert lib.GemmDesc.Ct.offset == 128 and lib.GemmDesc.n_trans_begin.offset == 152
    assert lib.GemmDesc.M.offset == 156 and lib.GemmDesc.epilogue.offset == 172 and lib.GemmDesc.tile_cfg.offset == 176
    assert lib.GemmDesc.row_stats_out.offset == 184 and lib.GemmDesc.ln_stats.offset == 208 and lib.GemmDesc.ln_colsum.offset == 232
    assert lib.GemmDesc.ln_inv_c.offset == 248 and lib.GemmDesc.ln_parts.offset == 256
    assert lib.GemmDesc.col_stats_out.offset == 264 and lib.GemmDesc.w_period.offset == 272 and ctypes.sizeof(lib.GemmDesc) == 280
    assert lib.ConvDesc.B.offset == 48 and lib.ConvDesc.tile_cfg.offset == 72 and lib.ConvDesc.col_stats_out.offset == 80
    assert lib.ConvDesc.S1.offset == 88 and lib.ConvDesc.S2.offset == 96 and lib.ConvDesc.S2_channels.offset == 108 and ctypes.sizeof(lib.ConvDesc) == 112


def test_geometry_helper_without_gpu():
    from tweediemix_amd import lib
    l = lib.load()
    assert l.tmix_groupnorm_ws_chunks(16384) == 128 and l.tmix_groupnorm_ws_chunks(1024) == 128 and l.tmix_groupnorm_ws_chunks(16) == 2 and l.tmix_groupnorm_ws_chunks(4) == 1


def test_ops_refuse_cpu_tensors():
    import torch
    from tweediemix_amd import lib, ops
    with pytest.raises(lib.TmixError):
        ops.gemm(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))


def test_argument_errors_are_reported_before_any_launch():
    """every entry point validates its arguments first and returns a TMIX_E* code with a message (no GPU needed to see that)."""
    import ctypes as C
    from tweediemix_amd import lib
    l = lib.load()
    fake = C.c_void_p(0x1000)                   # aligned, never dereferenced: validation fails first
    cases = [
        (l.tmix_vpred_step(None, None, None, 0, 16, 1.0, 1.0, 0.0, 1.0, 0.0, None), "vpred_step"),
        (l.tmix_vpred_step(fake, fake, fake, 7, 16, 1.0, 1.0, 0.0, 1.0, 0.0, None), "dtype"),
        (l.tmix_frame_inject(fake, 0, 2, 1, 64, 1, 0.0, 0.0, None), "frames"),
        (l.tmix_temporal_attn(fake, 3 * 64, fake, 64, 1, 17, 4, 1, 0.125, None), "frames"),
        (l.tmix_temporal_attn(fake, 64, fake, 64, 1, 16, 4, 1, 0.125, None), "ld"),
        (l.tmix_softmax_rows_causal(fake, 128, fake, 128, 10, 128, 1.0, 3, None), "seq"),
        (l.tmix_softmax_rows_masked(fake, 128, fake, 128, 10, 128, 0, 1.0, None), "valid"),
        (l.tmix_conv_in(fake, fake, None, fake, 1, 5, 8, 8, 64, None), "Cin"),
    ]
    for rc, word in cases:
        assert rc < 0, word
    assert l.tmix_gemm_stats_parts(1280, 7) == 8 and l.tmix_gemm_stats_parts(1280, 1) == 10 and l.tmix_gemm_stats_parts(1280, 99) == -1
    bm, bn = C.c_int(), C.c_int()
    assert l.tmix_gemm_tile_shape(9, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (256, 128)
    assert l.tmix_gemm_tile_shape(12, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (128, 160)
    for cfg, shp in ((13, (64, 160)), (14, (256, 320)), (15, (32, 160))):
        assert l.tmix_gemm_tile_shape(cfg, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == shp
    for cfg, shp in ((16, (256, 256)), (17, (256, 128))):          # phase-offset mainloop tilings
        assert l.tmix_gemm_tile_shape(cfg, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == shp
    assert l.tmix_gemm_tile_shape(18, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (128, 160)
    for cfg in (19, 20, 21):                                         # 128x160 with one / two / four loader waves
        assert l.tmix_gemm_tile_shape(cfg, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (128, 160)
    assert l.tmix_gemm_tile_shape(22, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (256, 320)       # 256 x 320, phase-offset loop
    assert l.tmix_gemm_tile_shape(23, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (128, 160)       # 2 x 2 waves of 64 x 80
    assert l.tmix_gemm_tile_shape(24, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (256, 320)       # (reserved id: dev builds run 256 x 320 on persistent workgroups, the shipped library tiling 14)
    assert l.tmix_gemm_tile_shape(25, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (128, 160)       # (reserved id: dev builds add an L2 prefetcher wave to tiling 23)
    assert l.tmix_gemm_tile_shape(26, C.byref(bm), C.byref(bn)) == 0 and (bm.value, bn.value) == (128, 160)       # halo-patch convolution: 4 x 32 pixels x 160 channels
    assert l.tmix_gemm_tile_shape(27, C.byref(bm), C.byref(bn)) < 0 and l.tmix_gemm_stats_parts(1280, 14) == 4
    # the whole-step entry points and the timing hook validate before touching the device
    assert l.tmix_step_prologue(fake, fake, fake, fake, 1, 4, 6, None) < 0            # n % 4
    assert l.tmix_fused_tweedie_step_dev(fake, fake, 0, None, 0, fake, None, 3, 4, 64, 0, 4, 1, fake, None) < 0   # FUSION without masks
    assert l.tmix_fused_tweedie_step_dev(fake, fake, 0, fake, 0, fake, None, 3, 4, 64, 0, 3, 1, fake, None) < 0   # too few eps rows
    assert l.tmix_prof_begin(None, 4, 0) < 0 and l.tmix_prof_end() == 0
    d = lib.GemmDesc()
    assert l.tmix_gemm_bf16(C.byref(d), None) < 0 and b"null" in l.tmix_last_error_string()
    # tmix_gemm_fp8 + col_stats_out: compiled for the bf16 tilings only -- refused, not remapped onto a bf16 kernel over e4m3 bytes
    d = lib.GemmDesc()
    d.A, d.W, d.C, d.col_stats_out = 0x1000, 0x2000, 0x3000, 0x4000
    d.M, d.N, d.K, d.batch, d.lda, d.ldw, d.ldc, d.n_trans_begin = 256, 256, 256, 1, 256, 256, 256, -1
    assert l.tmix_gemm_fp8(C.byref(d), C.cast(C.c_void_p(0x5000), C.POINTER(C.c_uint8)), C.cast(C.c_void_p(0x6000), C.POINTER(C.c_uint8)), None) == lib.EINVAL
    assert b"col_stats_out" in l.tmix_last_error_string()
    # round 4 entry points: the e4m3 forms validate before touching the device
    assert l.tmix_attn_fwd_f8(fake, 64, 64, fake, 64, 64, fake, 80, 64 * 80, fake, 64, None, 64, 1, 1, 64, 77, 0.125, None) == lib.EINVAL          # no scale array
    assert l.tmix_attn_fwd_f8(fake, 64, 64, fake, 64, 64, fake, 80, 64 * 80, fake, 32, fake, 64, 1, 1, 64, 77, 0.125, None) == lib.EINVAL          # rows narrower than H * 64
    cd = lib.ConvDesc()
    cd.X, cd.Wt, cd.Y = 0x1000, 0x2000, 0x3000
    cd.B, cd.H, cd.W, cd.Cin, cd.Cout = 1, 8, 8, 64, 64
    assert l.tmix_conv3x3_nhwc_fp8(C.byref(cd), C.c_void_p(0x4000), C.c_void_p(0x5000), None) == lib.EINVAL and b"128" in l.tmix_last_error_string()   # Cin % 128
    cd.Cin = 128
    assert l.tmix_conv3x3_nhwc_fp8(C.byref(cd), None, C.c_void_p(0x5000), None) == lib.EINVAL
    assert l.tmix_groupnorm_nhwc_pre_f8(fake, 48, None, 0, fake, fake, fake, fake, fake, 1, 64, 3, 1e-5, 1, fake, 48, None, 0, None) == lib.ESHAPE     # C % 32
    assert l.tmix_groupnorm_nhwc_pre_f8(fake, 64, None, 0, fake, None, fake, fake, fake, 1, 64, 32, 1e-5, 1, fake, 64, None, 0, None) == lib.EINVAL    # no scales
    d = lib.GemmDesc()
    d.A, d.W, d.C, d.Ct = 0x1000, 0x2000, 0x3000, 0x4000
    d.M, d.N, d.K, d.batch, d.lda, d.ldw, d.ldc, d.ldct, d.n_trans_begin, d.epilogue = 256, 1280, 256, 1, 256, 256, 640, 256, -1, lib.EPI_GEGLU
    d.tile_cfg, d.reserved0 = 21, lib.F8_GEGLU_OUT
    u8 = lambda a: C.cast(C.c_void_p(a), C.POINTER(C.c_uint8))
    assert l.tmix_gemm_fp8(C.byref(d), u8(0x5000), u8(0x6000), None) == lib.EINVAL and b"tile_cfg" in l.tmix_last_error_string()     # 160-wide tiles cannot end MX blocks of the GEGLU output

    # round 5 host rules (pure functions of the shape, no device): the self-attention key split applies to partly filled last rounds only ...
    W = lambda r, parts: 4096 + r * parts * 4 * 9 * 64 * 16
    assert l.tmix_attn_split_ws_bytes(4, 20, 1024, 1024) == W(128, 4) and l.tmix_attn_split_ws_bytes(4, 10, 4096, 4096) == W(256, 2)      # SDXL attn1 at 1024^2, B = 4
    assert l.tmix_attn_split_ws_bytes(8, 20, 1024, 1024) == W(256, 2) and l.tmix_attn_split_ws_bytes(1, 20, 4096, 4096) == W(128, 4)
    for shape in [(2, 20, 1024, 1024), (16, 20, 1024, 1024), (4, 20, 1024, 77), (4, 20, 1024, 1000), (1, 1, 128, 128), (4, 20, 1024, 64), (0, 20, 1024, 1024)]:
        assert l.tmix_attn_split_ws_bytes(*shape) == 0, shape
    assert l.tmix_attn_fwd_ws(fake, 1280, 1280 * 1024, fake, 1280, 1280 * 1024, fake, 1024, 1280 * 1024, None, 1280, 1280 * 1024, 4, 20, 1024, 1024, 0.125, fake, 1 << 30, None) == lib.EINVAL
    # ... and tmix_groupnorm_nhwc takes the one-launch form by the IMAGE's shape alone (a few groups' slice of at most 64 K elements)
    for (hw, c), n in {(336, 1280): 1, (84, 1280): 1, (336, 1920): 1, (1344, 320): 1, (100, 64): 1, (84, 2560): 1, (5376, 320): 3, (1024, 2560): 3, (16384, 320): 3, (1024, 1280): 1,
                       (4096, 640): 3}.items():
        assert l.tmix_groupnorm_nhwc_launches(hw, c, 32) == n, (hw, c)
    assert l.tmix_groupnorm_nhwc_launches(0, 320, 32) == 0 and l.tmix_groupnorm_nhwc_launches(64, 100, 32) == 0

    # the once-per-video conditioning kernels of the I2VGen-XL path (csrc/conditioning.hip)
    assert l.tmix_conv3x3_f32(fake, fake, None, fake, 1, 4, 8, 8, 16, 3, 0, None) == lib.ESHAPE and b"stride" in l.tmix_last_error_string()
    assert l.tmix_conv3x3_f32(fake, None, None, fake, 1, 4, 8, 8, 16, 1, 0, None) == lib.EINVAL
    assert l.tmix_adaptive_avgpool_f32(fake, fake, 4, 8, 8, 0, 4, None) == lib.ESHAPE
    assert l.tmix_linear_f32(fake, fake, None, fake, 257, 8, 8, 0, 0, None) == lib.ESHAPE
    assert l.tmix_i2v_temporal_encoder(fake, fake, 1, 16, 8, 64, *([fake] * 11), None) == lib.ESHAPE and b"channels" in l.tmix_last_error_string()
    assert l.tmix_i2v_temporal_encoder(fake, fake, 1, 17, 4, 64, *([fake] * 11), None) == lib.ESHAPE and b"frames" in l.tmix_last_error_string()
    assert l.tmix_i2v_temporal_encoder(fake, fake, 1, 16, 4, 64, *([fake] * 10), None, None) == lib.EINVAL
    # periodic weight sets: the period must divide the batch
    d = lib.GemmDesc()
    d.A, d.W, d.C = 0x1000, 0x2000, 0x3000
    d.M, d.N, d.K, d.batch, d.lda, d.ldw, d.ldc, d.n_trans_begin = 64, 64, 64, 6, 64, 64, 64, -1
    d.strideA, d.strideW, d.strideC, d.w_period = 64 * 64, 64 * 64, 64 * 64, 4
    assert l.tmix_gemm_bf16(C.byref(d), None) == lib.ESHAPE and b"w_period" in l.tmix_last_error_string()
