// gemm_conv.hip -- host entry points of the bf16 / fp8 MFMA GEMM and the 3x3 NHWC implicit-GEMM convolution: argument
// validation, descriptor -> kernel parameters, tiling switch.  The kernel lives in gemm_kernel.h and is instantiated per group
// of tilings in gemm_inst_*.hip.
#include "gemm_kernel.h"
#include <stdlib.h>

using namespace tmix_gemm;

namespace {

int pick_cfg(const Params& p, int batch) {
    // heuristic default (the plan builder can override per shape after timing the candidates):
    // 256x128 when it still yields >= 3/4 of a CU wave, else 128x128.
    const int64_t t2 = (int64_t)((p.M + 255) / 256) * ((p.N + 127) / 128) * batch;
    return t2 >= 192 ? 2 : 1;
}

int launch(int conv, Params& p, int batch, int cfg, hipStream_t st) {
    if (cfg <= 0 || cfg > NUM_CFG) cfg = pick_cfg(p, batch);
    if (cfg == 8) cfg = 19; else if (cfg == 9) cfg = 2; else if (cfg == 10) cfg = 1; else if (cfg == 11) cfg = 4;   // round-1 loader-wave tilings, retired (3 waves / SIMD register budget: they spilled)
    // 23 = 128x160 over 2 x 2 math waves of 64x80 (v_mfma_f32_16x16x32_bf16) + four loader waves (gemm_w22.hip): the staged plain bf16 epilogue only
    // 26 = the halo-patch convolution (gemm_convh.hip): stride 1, bf16, 4 x 32 pixel tiles; anything it does not carry runs as the loader-wave tilings 20 (conv) / 21 (GEMM)
    if (cfg == 26) { if (convh_eligible(p, conv, p.scaleA != nullptr)) return launch_convh(p, st); cfg = conv ? 20 : 21; }
#ifdef TMIX_EXPERIMENTAL_TILINGS      // dev variants (make EXPERIMENTAL=1): 24 = 256x320 on persistent workgroups (gemm_ff1p.hip), 25 = tiling 23 with an L2 prefetcher wave,
    // 27 (asked for as tile_cfg 27 through the 13 slot: TMIX_TILE13_NS2) = 64x160 over five waves with a TWO-deep ring, two workgroups per CU (VERDICT r5 item 6 i)
    if (cfg == 23 || cfg == 25) { if (w22_eligible(p, conv, 0)) return launch_w22(p, batch, st, cfg == 25); cfg = conv ? 12 : 21; }
    if (cfg == 24) { if (ff1p_eligible(p, conv, 0, batch)) return launch_ff1p(p, st); cfg = 14; }
#else                                 // the shipped library: ids 24 / 25 are reserved and run as the tilings they were variants of (same bits)
    if (cfg == 24) cfg = 14; else if (cfg == 25) cfg = 23;
    if (cfg == 23) { if (w22_eligible(p, conv, 0)) return launch_w22(p, batch, st, 0); cfg = conv ? 12 : 21; }
#endif
    if (cfg == 6) cfg = 4;      // 256x256 over four waves (128x128 wave tiles) is retired: it spilled and lost everywhere; same tile shape over eight waves
    if (p.n_trans_begin >= 0) {
        int bm = 0, bn = 0;
        tmix_gemm_tile_shape(cfg, &bm, &bn);
        if (p.n_trans_begin % bn) cfg = 2;                                     // the boundary must fall on a tile edge (N = 3 x 320: 640)
        if (!(p.wide & 4)) {                                                   // narrow (unstaged) transposed stores need square wave tiles
            if (cfg == 4 || cfg == 5 || cfg == 7 || cfg >= 12) cfg = 2;      // (18 included)
        }
    }
    if (p.f8copy) {                // the e4m3 copy of C is compiled into the tilings that have registers to spare for it (F8C)
        static const int alt[NUM_CFG + 1] = {0, 1, 2, 3, 4, 5, 4, 7, 7, 2, 1, 4, 12, 13, 12, 15, 16, 17, 18, 12, 12, 12, 12, 12, 12, 12, 12};
        // (every substitute keeps the tile width, and with it the number of row-statistics partials, except 256x320 -> 128x160)
        if (cfg == 14 && p.stats_out) TMIX_FAIL(TMIX_EINVAL, "gemm: the e4m3 copy is not compiled into tiling 14; with row_stats_out pick another tiling (the partial count depends on it)");
        if (!((cfg == 19 || cfg == 20 || cfg == 21) && !conv && p.scaleA && p.K % 128 == 0)) cfg = alt[cfg];      // (the loader-wave tilings on e4m3 operands carry the copy themselves)
    }
    int f8 = 0;
    if (!conv && p.scaleA) {     // fp8 operands (tmix_gemm_fp8): the phase-offset loop only; 256x128 tiles for narrow N
        f8 = p.ldScaleA ? 2 : 1;
        const int asked = cfg;
        // the lock-step loops on e4m3 operands (128 x 160 with / without loader waves, 256 x 320): rows of 128 K values
        // (not for the e4m3 GEGLU output: its MX blocks of 32 output columns need wave tiles that are multiples of 64 weight rows wide -- the
        // phase-offset tilings' 128 x 64; a 160-wide tile ends in the middle of a block)
        if ((cfg == 12 || cfg == 19 || cfg == 20 || cfg == 21) && p.K % 128 == 0 && !p.f8out) {
            f8 += 2;
            // the loader-wave instantiation at the 256-register limit of two waves per SIMD holds the staged plain / GEGLU epilogues only (the transposed and
            // narrow forms spill there, and scratch traffic would break the counted vmcnt): those launches run without loader waves
            const bool staged = p.n_trans_begin < 0 && (p.epilogue == TMIX_EPI_GEGLU ? (p.wide & 2) : (p.wide & 1));
            const bool lw = cfg != 12;
            if (lw && !staged) cfg = 12;
            // ... and the e4m3 copy of C in its straight-line form only (epilogue family 4: bf16 output, no activation, no row-group bias)
            if (lw && p.f8copy && (p.epilogue != TMIX_EPI_NONE || p.rgb)) cfg = 12;
        } else
        cfg = (cfg == 17 || (cfg != 16 && (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256) * batch < 160)) ? 17 : 16;
        if (f8 == 2 && cfg == 16 && p.K / 32 > f8_block_cap(256)) cfg = 17;    // the tile's block scales stay in LDS beside the ring
        // the consumers of row_stats_out were told the partial count of the REQUESTED tiling (tmix_gemm_stats_parts): never change the width under them
        int abm = 0, abn = 0, cbm = 0, cbn = 0;
        tmix_gemm_tile_shape(asked, &abm, &abn); tmix_gemm_tile_shape(cfg, &cbm, &cbn);
        if (p.stats_out && abn != cbn) TMIX_FAIL(TMIX_EINVAL, "gemm_fp8: tile_cfg %d cannot run this launch (K / 32 = %d block scales per row exceed the 256x256 tile's LDS budget); "
                                                                "with row_stats_out request tile_cfg 17 explicitly", asked, p.K / 32);
    }
    if (conv && p.scaleA) {     // convolution on e4m3 operands: the 128 x 160 lock-step tilings, with two loader waves (20) or without (12)
        f8 = 4;
        cfg = (cfg == 20 || cfg == 21 || cfg == 19) ? 20 : 12;
    } else
    if (conv) {
        // the phase-offset and loader-wave mainloops exist for the plain GEMM only (the im2col gather's per-row offset tables do
        // not fit a loader wave's register budget, and the conv mainloop already runs at 0.8-1.0 PFLOP/s): nearest plain tiling
        // (tiling 20 -- 128 x 160 plus TWO loader waves -- also exists for the convolution since the kernels are instantiated per epilogue family:
        // 244 VGPRs, no scratch; not with shortcut taps, whose source switch lives in the staging path of the math waves' kernel)
        if (cfg == 16) cfg = 4;
        else if (cfg == 17) cfg = 2;
        else if (cfg == 22) cfg = 14;
        else if (cfg >= 18 && cfg != 20) cfg = 12;
    } else if (cfg == 16 && !f8 && (p.K % 32)) cfg = 4;
    // 256x320 phase-offset: bf16, staged GEGLU / plain epilogues only (the transposed and narrow forms spill at its register count)
    if (!conv && cfg == 22 && (f8 || p.n_trans_begin >= 0 || !(p.epilogue == TMIX_EPI_GEGLU ? (p.wide & 2) : (p.wide & 1)))) cfg = f8 ? 16 : 14;
    // column statistics (cs_out) are compiled for the lock-step tilings only: the phase-offset ones run as their nearest plain tiling
    if (p.cs_out) { if (cfg == 16) cfg = 4; else if (cfg == 17) cfg = 2; else if (cfg == 22) cfg = 14; }
    int rc = f8 >= 3 ? -999 : launch_group0(cfg, conv, f8, p, batch, st);
    if (rc == -999 && f8 < 3) rc = launch_group1(cfg, conv, f8, p, batch, st);
    if (rc == -999 && f8 < 3) rc = launch_group2(cfg, conv, f8, p, batch, st);
    if (rc == -999 && f8 < 3) rc = launch_group3(cfg, conv, f8, p, batch, st);
    if (rc == -999) rc = launch_group4(cfg, conv, f8, p, batch, st);
    if (rc == -999) rc = launch_group5(cfg, conv, f8, p, batch, st);
    if (rc == -999) TMIX_FAIL(TMIX_EINVAL, "gemm: no kernel for tile_cfg %d", cfg);
    return rc;
}

}  // namespace

extern "C" int tmix_gemm_tile_shape(int tile_cfg, int* bm, int* bn) {
    static const int shape[NUM_CFG + 1][2] = {{0, 0}, {128, 128}, {256, 128}, {128, 128}, {256, 256}, {256, 128}, {256, 256}, {128, 160},
                                              {128, 160}, {256, 128}, {128, 128}, {256, 256}, {128, 160}, {64, 160}, {256, 320}, {32, 160},
                                              {256, 256}, {256, 128}, {128, 160}, {128, 160}, {128, 160}, {128, 160}, {256, 320}, {128, 160}, {256, 320}, {128, 160}, {128, 160}};
    if (tile_cfg < 1 || tile_cfg > NUM_CFG || !bm || !bn) TMIX_FAIL(TMIX_EINVAL, "gemm_tile_shape: tile_cfg=%d", tile_cfg);
    *bm = shape[tile_cfg][0]; *bn = shape[tile_cfg][1];
    return TMIX_OK;
}

extern "C" int tmix_gemm_stats_parts(int N, int tile_cfg) {
    int bm, bn;
    if (N <= 0 || tmix_gemm_tile_shape(tile_cfg, &bm, &bn) != TMIX_OK) return -1;
    return (N + bn - 1) / bn;
}

// fp8 = true: A / W hold OCP e4m3 bytes with one E8M0 scale per row (tmix_gemm_fp8); everything behind the main loop is shared
static int gemm_entry(const tmix_gemm_desc* d, bool fp8, const uint8_t* scaleA, const uint8_t* scaleW, void* stream, const QAExtra* qa = nullptr) {
    if (!d || !d->A || !d->W) TMIX_FAIL(TMIX_EINVAL, "gemm: null descriptor/operand");
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) TMIX_FAIL(TMIX_ESHAPE, "gemm: empty problem M=%d N=%d K=%d batch=%d", d->M, d->N, d->K, d->batch);
    if (d->K % BK) TMIX_FAIL(TMIX_ESHAPE, "gemm: K=%d must be a multiple of %d", d->K, BK);
    if (d->N % 4) TMIX_FAIL(TMIX_ESHAPE, "gemm: N=%d must be a multiple of 4", d->N);
    const int el = fp8 ? 1 : 2, al = 16 / el;            // element bytes, elements per 16 bytes
    if (fp8 && (!scaleA || !scaleW)) TMIX_FAIL(TMIX_EINVAL, "gemm_fp8: null scale vector");
    if ((d->lda % al) || (d->ldw % al) || (d->strideA % al) || (d->strideW % al)) TMIX_FAIL(TMIX_EALIGN, "gemm: lda/ldw/strides must be multiples of %d elements", al);
    if (!aligned16(d->A) || !aligned16(d->W)) TMIX_FAIL(TMIX_EALIGN, "gemm: A/W must be 16-byte aligned");
    const bool has_trans = d->n_trans_begin >= 0 && d->n_trans_begin < d->N;
    if (has_trans && (!d->Ct || (d->n_trans_begin % 128))) TMIX_FAIL(TMIX_EINVAL, "gemm: transposed region needs Ct and n_trans_begin %% 128 == 0");
    if ((!has_trans || d->n_trans_begin > 0) && !d->C && !qa) TMIX_FAIL(TMIX_EINVAL, "gemm: null C");
    if (d->C && ((d->ldc % 4) || (((uintptr_t)d->C) & 7) || (d->strideC % 4))) TMIX_FAIL(TMIX_EALIGN, "gemm: C must be 8-byte aligned with ldc %% 4 == 0");
    if (d->residual && ((d->ldr % 4) || (((uintptr_t)d->residual) & 7))) TMIX_FAIL(TMIX_EALIGN, "gemm: residual alignment");
    if (d->bias && (((uintptr_t)d->bias) & 15)) TMIX_FAIL(TMIX_EALIGN, "gemm: bias must be 16-byte aligned");
    if (d->rowgroup_bias && (d->rows_per_group <= 0 || (((uintptr_t)d->rowgroup_bias) & 15))) TMIX_FAIL(TMIX_EINVAL, "gemm: rowgroup_bias needs rows_per_group > 0 and 16-byte alignment");
    if (d->epilogue == TMIX_EPI_GEGLU && ((d->N % 32) || has_trans || d->residual || d->rowgroup_bias)) TMIX_FAIL(TMIX_EINVAL, "gemm: GEGLU needs N %% 32 == 0 and no residual/transposed region");
    if (d->epilogue < TMIX_EPI_NONE || d->epilogue > TMIX_EPI_QUICKGELU) TMIX_FAIL(TMIX_EINVAL, "gemm: bad epilogue %d", d->epilogue);
    if ((d->epilogue == TMIX_EPI_GELU || d->epilogue == TMIX_EPI_QUICKGELU) && (has_trans || d->residual)) TMIX_FAIL(TMIX_EINVAL, "gemm: activation epilogues take no residual/transposed region");
    if (d->epilogue == TMIX_EPI_F32OUT && (has_trans || (((uintptr_t)d->C) & 15))) TMIX_FAIL(TMIX_EINVAL, "gemm: fp32 output needs a 16-byte aligned C and no transposed region");
    if ((int64_t)d->M * d->lda >= (1ll << 31) || (int64_t)d->N * d->ldw >= (1ll << 31)) TMIX_FAIL(TMIX_ESHAPE, "gemm: operand extent exceeds 32-bit element offsets");
    if (d->w_period < 0 || (d->w_period > 0 && (d->batch % d->w_period))) TMIX_FAIL(TMIX_ESHAPE, "gemm: w_period=%d must divide batch=%d", d->w_period, d->batch);
    Params p = {};
    p.A = (const bf16_t*)d->A; p.lda = d->lda; p.strideA = d->strideA;
    // (w_period == batch is "every slice its own set": the plain strideW walk -- one group would make the magic divisor wrap to 1)
    const int wper = (d->w_period > 0 && d->w_period < d->batch) ? d->w_period : 0;
    p.w_period = wper; p.w_groups = wper ? d->batch / wper : 1;
    p.w_magic = (unsigned)((1ull << 32) / (unsigned)p.w_groups) + 1u;
    if (d->batch > 65535) TMIX_FAIL(TMIX_ESHAPE, "gemm: batch=%d exceeds the grid's 65535 slices", d->batch);
    p.W = (const bf16_t*)d->W; p.ldw = d->ldw; p.strideW = d->strideW;
    p.C = (bf16_t*)d->C; p.ldc = d->ldc; p.strideC = d->strideC;
    p.bias = d->bias; p.strideBias = d->strideBias;
    p.R = (const bf16_t*)d->residual; p.ldr = d->ldr; p.strideR = d->strideR;
    p.rgb = d->rowgroup_bias; p.rows_per_group = d->rows_per_group;
    p.Ct = (bf16_t*)d->Ct; p.ldct = d->ldct; p.strideCt = d->strideCt;
    p.n_trans_begin = has_trans ? d->n_trans_begin : -1;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.epilogue = d->epilogue;
    p.bytesA = (unsigned)(((int64_t)(d->M - 1) * d->lda + d->K) * el);
    p.bytesW = (unsigned)(((int64_t)(d->N - 1) * d->ldw + d->K) * el);
    if (fp8) {
        const bool lockstep = (d->tile_cfg == 12 || (d->tile_cfg >= 19 && d->tile_cfg <= 21)) && d->K % 128 == 0 && !(d->reserved0 & TMIX_F8_GEGLU_OUT);
        if (d->tile_cfg != TMIX_TILE_AUTO && d->tile_cfg != 16 && d->tile_cfg != 17 && !lockstep)
            TMIX_FAIL(TMIX_EINVAL, "gemm_fp8: tile_cfg must be AUTO, 16 (256x256), 17 (256x128) or -- K %% 128 == 0, no e4m3 GEGLU output -- 12 / 19 / 20 / 21 (128x160 without / with one, two, four loader waves)");
        p.scaleA = scaleA; p.scaleW = scaleW; p.strideScaleA = d->strideA ? d->M : 0; p.strideScaleW = d->strideW ? d->N : 0;
        if (d->reserved0 & TMIX_F8_A_BLOCK_SCALES) {
            p.ldScaleA = (int64_t)d->batch * d->M;                                                // [K/32][batch * M], dense
            if ((d->M % 4) || (((uintptr_t)scaleA) & 3)) TMIX_FAIL(TMIX_EALIGN, "gemm_fp8: block-scaled A needs M %% 4 == 0 and a 4-byte aligned scale array");
            if (!lockstep && d->K / 32 > f8_block_cap(128)) TMIX_FAIL(TMIX_ESHAPE, "gemm_fp8: block-scaled A supports K <= %d", 32 * f8_block_cap(128));
            if ((int64_t)(d->K / 32) * p.ldScaleA >= (1ll << 31)) TMIX_FAIL(TMIX_ESHAPE, "gemm_fp8: block scale array exceeds 32-bit offsets");
        }
        if (d->reserved0 & TMIX_F8_GEGLU_OUT) {
            if (d->epilogue != TMIX_EPI_GEGLU || !d->Ct || d->ldct < (int64_t)d->batch * d->M || (d->ldc % 8) || (d->strideC % 8) || (((uintptr_t)d->C) & 7))
                TMIX_FAIL(TMIX_EINVAL, "gemm_fp8: e4m3 GEGLU output needs the GEGLU epilogue, Ct = scale buffer [N/64][ldct >= batch*M] and 8-byte aligned C rows");
            p.f8out = 1; p.scale_out = (unsigned char*)d->Ct; p.ldScaleOut = d->ldct;
        }
    } else if (d->reserved0 & ~TMIX_F8_COPY_OUT) TMIX_FAIL(TMIX_EINVAL, "gemm: the fp8 operand flags in reserved0 belong to tmix_gemm_fp8");
    if (d->reserved0 & TMIX_F8_COPY_OUT) {
        if (has_trans || d->epilogue != TMIX_EPI_NONE || (d->reserved0 & TMIX_F8_GEGLU_OUT) || !d->Ct || (d->M % 32) || (d->N % 32) || (d->ldct % 8) || d->ldct < d->N
            || (((uintptr_t)d->Ct) & 7) || d->strideCt < (int64_t)d->batch * d->M * d->ldct)
            TMIX_FAIL(TMIX_EINVAL, "gemm: the e4m3 copy needs the plain bf16 epilogue, M %% 32 == 0, N %% 32 == 0, Ct = bytes [batch*M][ldct >= N, %% 8 == 0] and strideCt = byte offset of the scale array behind them");
        p.f8copy = (unsigned char*)d->Ct; p.ldF8copy = d->ldct;
        p.scale_out = (unsigned char*)d->Ct + d->strideCt; p.ldScaleOut = (int64_t)d->batch * d->M;
    }
    if (d->row_stats_out && (has_trans || d->epilogue != TMIX_EPI_NONE)) TMIX_FAIL(TMIX_EINVAL, "gemm: row_stats_out needs the plain bf16 epilogue");
    if (d->row_stats_out && (d->tile_cfg < 1 || d->tile_cfg > NUM_CFG || (((uintptr_t)d->row_stats_out) & 7) || (d->strideStatsOut & 1) || d->ldStatsOut < d->M))
        TMIX_FAIL(TMIX_EINVAL, "gemm: row_stats_out needs an explicit tile_cfg (its partial count depends on it) and 8-byte alignment");
    if (d->ln_stats && (!d->ln_colsum || !(d->ln_inv_c > 0.f) || d->ln_parts < 1 || d->ln_parts > 16 || (int64_t)d->ln_parts * d->ldLnStats * 8 >= (1ll << 31) || (((uintptr_t)d->ln_stats) & 7) || (d->strideLnStats & 1) || d->ldLnStats < d->M))
        TMIX_FAIL(TMIX_EINVAL, "gemm: fused LayerNorm needs ln_colsum, ln_inv_c > 0, 1 <= ln_parts <= 16 and 8-byte aligned ln_stats (< 2 GiB)");
    p.stats_out = d->row_stats_out; p.strideStatsOut = d->strideStatsOut; p.ldStatsOut = d->ldStatsOut;
    p.ln_stats = d->ln_stats; p.strideLnStats = d->strideLnStats; p.ldLnStats = d->ldLnStats; p.ln_parts = d->ln_parts;
    p.ln_colsum = d->ln_colsum; p.strideLnColsum = d->strideLnColsum; p.ln_inv_c = d->ln_inv_c; p.ln_eps = d->ln_eps;
    // LDS-staged 16-byte stores need 16-byte aligned rows (the narrow 8-byte form stays as the fallback for odd strides)
    const bool c16 = d->C && aligned16(d->C) && (d->ldc % 8) == 0 && (d->strideC % 8) == 0;
    const bool r16 = !d->residual || (aligned16(d->residual) && (d->ldr % 8) == 0 && (d->strideR % 8) == 0);
    p.wide = 0;
    if (!tmix_env(TMIX_ENV_NARROW_EPILOGUE)) {
        if (d->epilogue == TMIX_EPI_GEGLU) { if (c16 || p.f8out) p.wide |= 2; }
        else if (d->epilogue == TMIX_EPI_F32OUT) { if (aligned16(d->C) && (d->ldc % 4) == 0 && (d->strideC % 4) == 0 && (d->N % 8) == 0) p.wide |= 1; }
        else if (c16 && r16 && (d->N % 8) == 0) p.wide |= 1;
        if (has_trans && aligned16(d->Ct) && (d->ldct % 8) == 0 && (d->strideCt % 8) == 0) p.wide |= 4;
    }
    if (d->col_stats_out) {
        if (fp8) TMIX_FAIL(TMIX_EINVAL, "gemm_fp8: col_stats_out is compiled for the bf16 tilings only (the fp8 loop would be remapped to a bf16 kernel over e4m3 bytes)");
        if (has_trans || d->epilogue != TMIX_EPI_NONE || d->row_stats_out || (d->reserved0 & TMIX_F8_COPY_OUT) || d->batch != 1 || (d->M % TMIX_COLSTATS_ROWS) || (d->N % 8)
            || !aligned16(d->col_stats_out) || !(p.wide & 1))
            TMIX_FAIL(TMIX_EINVAL, "gemm: col_stats_out needs the plain staged bf16 epilogue (16-byte aligned C / residual rows, N %% 8 == 0), batch == 1, M %% 32 == 0, "
                                   "and neither row_stats_out nor the e4m3 copy");
        p.cs_out = d->col_stats_out;
    }
    if (p.f8out && !(p.wide & 2)) TMIX_FAIL(TMIX_EINVAL, "gemm_fp8: e4m3 GEGLU output needs the staged epilogue");
    if (p.f8copy && !(p.wide & 1)) TMIX_FAIL(TMIX_EALIGN, "gemm: the e4m3 copy needs the staged epilogue (16-byte aligned C / residual rows, N %% 8 == 0)");
    if (fp8 && has_trans && !(p.wide & 4)) TMIX_FAIL(TMIX_EALIGN, "gemm_fp8: the transposed region needs a 16-byte aligned Ct with ldct %% 8 == 0");
    if (qa) return launch_qattn(p, *qa, d->batch, (hipStream_t)stream);
    return launch(0, p, d->batch, d->tile_cfg, (hipStream_t)stream);
}

// attn2.to_q (the projection described by d: A = hidden state rows, W = to_q weight, bias / folded LayerNorm as in tmix_gemm_bf16; d->C is not written) and
// the cross-attention of its output against cached K [images][Skv][ldk] / V^T [images][N][ldvt = 80] in one launch: O [batch * M][ldo] receives
// softmax(q K^T * scale) V per 64-wide head.  rows_per_image: consecutive rows of A that share an image's keys (the latent's token count).
extern "C" int tmix_gemm_q_cross_attn(const tmix_gemm_desc* d, const void* K, int64_t ldk, int64_t strideK, const void* Vt, int64_t ldvt, int64_t strideVt,
                                      void* O, int64_t ldo, int rows_per_image, int Skv, float scale, void* stream) {
    if (!d) TMIX_FAIL(TMIX_EINVAL, "gemm_q_cross_attn: null descriptor");
    if (d->reserved0 || d->residual || d->row_stats_out || d->col_stats_out || d->rowgroup_bias || d->epilogue != TMIX_EPI_NONE || (d->n_trans_begin >= 0 && d->n_trans_begin < d->N))
        TMIX_FAIL(TMIX_EINVAL, "gemm_q_cross_attn: the projection takes a bias and a folded LayerNorm only");
    const QAExtra x = {K, ldk, strideK, Vt, ldvt, strideVt, O, ldo, rows_per_image, Skv, scale};
    return gemm_entry(d, false, nullptr, nullptr, stream, &x);
}

extern "C" int tmix_gemm_bf16(const tmix_gemm_desc* d, void* stream) { return gemm_entry(d, false, nullptr, nullptr, stream); }

extern "C" int tmix_gemm_fp8(const tmix_gemm_desc* d, const uint8_t* scale_a, const uint8_t* scale_w, void* stream) {
    return gemm_entry(d, true, scale_a, scale_w, stream);
}

static int conv_entry(const tmix_conv_desc* d, const uint8_t* scale_x, const uint8_t* scale_w, void* stream) {
    const bool fp8 = scale_x != nullptr;
    if (!d || !d->X || !d->Wt || !d->Y) TMIX_FAIL(TMIX_EINVAL, "conv3x3: null descriptor/operand");
    if (fp8 && (!scale_w || (d->Cin % 128) || d->S1 || d->S2 || (((uintptr_t)scale_x) & 3)))
        TMIX_FAIL(TMIX_EINVAL, "conv3x3_fp8: needs both scale arrays (scale_x 4-byte aligned), Cin %% 128 == 0 and no shortcut taps");
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: empty problem");
    if (d->Cin % BK) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: Cin=%d must be a multiple of %d", d->Cin, BK);
    if (d->Cout % 4) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: Cout=%d must be a multiple of 4", d->Cout);
    if (d->mode < TMIX_CONV_S1 || d->mode > TMIX_CONV_S2A) TMIX_FAIL(TMIX_EINVAL, "conv3x3: bad mode %d", d->mode);
    if ((d->mode == TMIX_CONV_S2 || d->mode == TMIX_CONV_S2A) && ((d->H | d->W) & 1)) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: stride-2 needs even H,W");
    if (!aligned16(d->X) || !aligned16(d->Wt) || (((uintptr_t)d->Y) & 7)) TMIX_FAIL(TMIX_EALIGN, "conv3x3: pointer alignment");
    if ((d->bias && (((uintptr_t)d->bias) & 15)) || (d->batch_bias && (((uintptr_t)d->batch_bias) & 15))) TMIX_FAIL(TMIX_EALIGN, "conv3x3: bias alignment");
    Params p = {};
    p.H = d->H; p.Wd = d->W; p.Cin = d->Cin; p.mode = d->mode;
    p.ntaps = d->mode == TMIX_CONV_T3 ? 3 : 9;
    const bool half = d->mode == TMIX_CONV_S2 || d->mode == TMIX_CONV_S2A;
    p.Ho = half ? d->H / 2 : (d->mode == TMIX_CONV_UP2 ? d->H * 2 : d->H);
    p.Wo = half ? d->W / 2 : (d->mode == TMIX_CONV_UP2 ? d->W * 2 : d->W);
    const int64_t M = (int64_t)d->B * p.Ho * p.Wo;
    if (M > 0x7fffffff / 4) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: too many output pixels");
    if ((int64_t)d->B * d->H * d->W * d->Cin >= (1ll << 30) || (int64_t)d->Cout * 9 * d->Cin >= (1ll << 30)) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: operand larger than 2 GiB");
    p.A = (const bf16_t*)d->X;
    p.W = (const bf16_t*)d->Wt; p.ldw = p.ntaps * (int64_t)d->Cin;
    p.C = (bf16_t*)d->Y; p.ldc = d->Cout;
    p.bias = d->bias;
    p.R = (const bf16_t*)d->residual; p.ldr = d->Cout;
    p.rgb = d->batch_bias; p.rows_per_group = p.Ho * p.Wo * (d->batch_bias_images > 1 ? d->batch_bias_images : 1);
    p.n_trans_begin = -1;
    p.M = (int)M; p.N = d->Cout; p.K = p.ntaps * d->Cin;
    if (d->S1 || d->S2) {            // shortcut taps behind the nine conv taps
        const int c1 = d->S1_channels, c2 = d->S2 ? d->S2_channels : 0;
        if (!d->S1 || d->mode != TMIX_CONV_S1 || c1 <= 0 || (c1 % BK) || c2 < 0 || (c2 % BK) || (d->S2 && c2 == 0) || !aligned16(d->S1) || (d->S2 && !aligned16(d->S2)) || d->residual)
            TMIX_FAIL(TMIX_EINVAL, "conv3x3: shortcut taps need the stride-1 mode, S1 (S2 only with S1), channel counts that are multiples of %d, 16-byte alignment and no residual", BK);
        if ((int64_t)d->B * d->H * d->W * (c1 > c2 ? c1 : c2) >= (1ll << 30) || (int64_t)d->Cout * (9ll * d->Cin + c1 + c2) >= (1ll << 30)) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: shortcut operand larger than 2 GiB");
        p.S1 = (const bf16_t*)d->S1; p.S2 = (const bf16_t*)d->S2; p.c1s = c1; p.c2s = c2;
        p.bytesS1 = (unsigned)((int64_t)d->B * d->H * d->W * c1 * 2); p.bytesS2 = (unsigned)((int64_t)d->B * d->H * d->W * c2 * 2);
        p.K += c1 + c2;
        p.ldw = p.K;
    }
    p.epilogue = TMIX_EPI_NONE;
    p.bytesA = (unsigned)((int64_t)d->B * d->H * d->W * d->Cin * (fp8 ? 1 : 2));
    p.bytesW = (unsigned)((int64_t)d->Cout * p.K * (fp8 ? 1 : 2));
    if (fp8) { p.scaleA = scale_x; p.scaleW = scale_w; p.ldScaleA = d->Cin / 32; }
    p.wide = (!tmix_env(TMIX_ENV_NARROW_EPILOGUE) && aligned16(d->Y) && (d->Cout % 8) == 0 && (!d->residual || aligned16(d->residual))) ? 1 : 0;
    if (p.S1 && !p.wide) TMIX_FAIL(TMIX_EALIGN, "conv3x3: shortcut taps need the staged epilogue (16-byte aligned Y, Cout %% 8 == 0)");
    if (d->col_stats_out) {
        if ((M % TMIX_COLSTATS_ROWS) || !aligned16(d->col_stats_out) || !p.wide)
            TMIX_FAIL(TMIX_EINVAL, "conv3x3: col_stats_out needs B*Ho*Wo %% 32 == 0, Cout %% 8 == 0 and 16-byte aligned Y / residual");
        p.cs_out = d->col_stats_out;
    }
    return launch(1, p, 1, d->tile_cfg, (hipStream_t)stream);
}

extern "C" int tmix_conv3x3_nhwc(const tmix_conv_desc* d, void* stream) { return conv_entry(d, nullptr, nullptr, stream); }

extern "C" int tmix_conv3x3_nhwc_fp8(const tmix_conv_desc* d, const uint8_t* scale_x, const uint8_t* scale_w, void* stream) {
    if (!scale_x) TMIX_FAIL(TMIX_EINVAL, "conv3x3_fp8: null scale array");
    return conv_entry(d, scale_x, scale_w, stream);
}
