// gemm_ff1p.hip -- the GEGLU up-projection (FeedForward net.0 of every BasicTransformerBlock: value * gelu(gate) of Linear(C, 8C)(LayerNorm(h)), behind
// fusion_sampling.py:340) on PERSISTENT workgroups (TMIX_TILE_256x320_P = 24).
//
// 4096 x 10240 x 1280 is 512 tiles of 256 x 320 -- two rounds on 256 CUs -- and 23 % of the step.  In the captured step a workgroup of tiling 14 spends
// 4.4 us in its prologue, 34.8 in its K loop and 5.0 in its GEGLU epilogue (tools/insitu_phases.py), two of them are 88 us, the launch takes 104: every
// CU pays a workgroup hand-over between the rounds and the second workgroup starts as cold as the first.  Here ONE workgroup per CU walks its tiles:
// K-tile 0 of the next tile is requested into the free ring slot under the last K-tile of the current one, its LayerNorm row statistics and bias behind the
// epilogue's stores, and the K loop of the next tile starts from operands that are already there.
// Tile, wave layout and arithmetic are tiling 14's: eight waves (4 x 2) of 64 x 160 on v_mfma_f32_32x32x16_bf16, 72 KB K-tiles through a two-slot ring by
// LDS-DMA (gemm_kernel.h's swizzle), W fragments single-buffered, LayerNorm folded in (one extra MFMA k-step + rstd in the epilogue), GEGLU on the
// accumulators (weight rows interleaved in 16-row value / gate groups), LDS-staged 16-byte stores.  Per-lane staging offsets do not depend on the tile
// (M % 256 == 0, N % 320 == 0 are required; the tile's origin rides in the scalar offset of the buffer loads).
//
// MFMA roofline: 2*M*N*K flops per launch against the 2.5 PFLOP/s dense bf16 peak.
#include "gemm_kernel.h"

namespace tmix_gemm {

namespace {

constexpr int F_BM = 256, F_BN = 320, F_NW = 8;
constexpr int F_TM = 64, F_TN = 160, F_FM = 2, F_FN = 5;
constexpr int F_ATILE = F_BM * 128, F_BTILE = F_BN * 128, F_STAGE = F_ATILE + F_BTILE, F_RING = 2 * F_STAGE;
constexpr int F_RA = (F_BM / 8) / F_NW, F_RB = (F_BN / 8) / F_NW, F_L = F_RA + F_RB;         // LDS-DMA instructions per wave and K-tile: 4 + 5
constexpr int F_STG = 32 * (64 * 2 + 16);             // a wave's output patch: 32 rows x 64 bf16 columns (+ pad)
static_assert(F_NW * F_STG <= F_STAGE, "the epilogue patches fit in one ring slot");

__global__ void __launch_bounds__(F_NW * 64, 2) gemm_ff1p_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef TMIX_NO_KERNARG_TOUCH
    kernarg_touch<(int)sizeof(Params)>();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int l31 = lane & 31, lhi = lane >> 5, lrow = lane >> 3;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, blockIdx.x == 0, p.prof_detail);
    // the NEXT launch's weights (tmix_gemm_prefetch_next): touched in front of the first K-tile
    // (ONE sink register for all touches -- in-order returns; "+v" keeps it live between them -- released behind the last tile: eight kept registers would stay
    // live across the whole tile loop, which has none to spare)
    unsigned pf_sink = 0;
    if (p.pf) {
        const long long nwg = gridDim.x, nth = F_NW * 64;
        const long long lines = (p.pf_bytes + 127) >> 7; const int per = p.pf_per;
        long long ln = (long long)blockIdx.x * nth + tid;
        for (int u = 0; u < per; ++u, ln += nwg * nth)
            if (ln < lines) asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(p.pf + (ln << 7)) : "memory");
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.bytesW, 0x00020000);
    // tile-independent per-lane staging offsets: instruction idx = 8 r + w covers LDS rows 8 idx .. 8 idx + 7; position q of row r holds source chunk
    // q ^ ((r >> 1) & 7) = q ^ ((4 idx + (lane >> 4)) & 7), which does not depend on r (32 r is a multiple of 8): ONE offset register per operand, the
    // instruction's 64-row advance rides in the scalar offset with the tile's origin and the K-tile
    const unsigned swz = (unsigned)(((lane & 7) ^ ((4 * w + (lane >> 4)) & 7)) * 16);
    const unsigned aoff = (unsigned)(w * 8 + lrow) * (unsigned)p.lda * 2u + swz, woff = (unsigned)(w * 8 + lrow) * (unsigned)p.ldw * 2u + swz;
    const unsigned a_adv = 64u * (unsigned)p.lda * 2u, w_adv = 64u * (unsigned)p.ldw * 2u;
    const int nk = p.K / BK;
    const int ntiles = p.tiles_m * p.tiles_n, nwg = gridDim.x;
    // virtual tile id v = blockIdx.x + round * gridDim.x -> logical id (XCD-contiguous, gemm_kernel.h's patches of group_m x tiles_n) -> (tile_m, tile_n)
    auto tile_origin = [&](int v, int& m0, int& n0) __attribute__((always_inline)) {
        const int bid = xcd_remap(v, ntiles);
        const int per_group = p.group_m * p.tiles_n;
        const int grp = bid / per_group;
        const int first_m = grp * p.group_m;
        const int gsize = min(p.tiles_m - first_m, p.group_m);
        const int rem = bid - grp * per_group;
        const int tn = rem / gsize, tm = first_m + (rem - tn * gsize);
        m0 = tm * F_BM; n0 = tn * F_BN;
    };
    auto dma_piece = [&](const int r, char* slot, unsigned sa, unsigned sw_) __attribute__((always_inline)) {
        if (r < F_RA) blds16(rsA, aoff, sa + (unsigned)r * a_adv, slot + (r * F_NW + w) * 1024);
        else blds16(rsW, woff, sw_ + (unsigned)(r - F_RA) * w_adv, slot + F_ATILE + ((r - F_RA) * F_NW + w) * 1024);
    };

    uint4* ln_mfrag = (uint4*)(smem + F_RING);
    uint4* ln_cfrag = ln_mfrag + F_BM;
    float* ln_rs = (float*)(ln_cfrag + F_BN);
    float* bias_lds = ln_rs + F_BM;
    const bool ln_on = p.ln_stats != nullptr;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_stats, 0, ln_on ? (int)(p.ln_parts * p.ldLnStats * 8) : 0, 0x00020000);
    constexpr int PU = 16;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

    const int fsw = (lane >> 1) & 7;
    const int offA = (wr * F_TM + l31) * 128, offW = F_ATILE + (wc * F_TN + l31) * 128;
    int v = blockIdx.x;
    if (v >= ntiles) return;
    int m0, n0;
    tile_origin(v, m0, n0);
    int cur = 0;
    // K-tile 0 of the first tile
#pragma unroll
    for (int r = 0; r < F_L; ++r) dma_piece(r, smem, (unsigned)m0 * (unsigned)p.lda * 2u, (unsigned)n0 * (unsigned)p.ldw * 2u);
    bool first_tile = true;
    for (;;) {
        // ---- the tile's LayerNorm row statistics / weight column sums / bias -> LDS (thread t owns tile row t and tile column t)
        {
            u32x2 lnv[PU];
            float ln_cs = 0.f, bias_r = 0.f;
            if (ln_on) {
                if (tid < F_BM) {
#pragma unroll
                    for (int q = 0; q < PU; ++q)
                        if (q < p.ln_parts) lnv[q] = __builtin_amdgcn_raw_buffer_load_b64(rsS, (m0 + tid) * 8, q * (int)p.ldLnStats * 8, 0);
                }
                if (tid < F_BN) ln_cs = p.ln_colsum[n0 + tid];
            }
            if (p.bias && tid < F_BN) bias_r = p.bias[n0 + tid];
            if (tid < F_BN) bias_lds[tid] = bias_r;
            if (ln_on) {
                if (tid < F_BM) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int q = 0; q < PU; ++q)
                        if (q < p.ln_parts) { s1 += __uint_as_float(lnv[q].x); s2 += __uint_as_float(lnv[q].y); }
                    const float mean = s1 * p.ln_inv_c;
                    ln_rs[tid] = rsqrtf(fmaxf(s2 * p.ln_inv_c - mean * mean, 0.f) + p.ln_eps);
                    const float x = -mean;
                    const unsigned x1 = __float_as_uint(x) & 0xffff0000u;
                    const float r1 = x - __uint_as_float(x1);
                    const unsigned x2 = __float_as_uint(r1) & 0xffff0000u;
                    const unsigned x3 = __float_as_uint(r1 - __uint_as_float(x2)) & 0xffff0000u;
                    ln_mfrag[tid] = make_uint4((x1 >> 16) | x1, x2 >> 16, (x1 >> 16) | x3, x2 >> 16);
                }
                if (tid < F_BN) {
                    const unsigned x1 = __float_as_uint(ln_cs) & 0xffff0000u;
                    const float r1 = ln_cs - __uint_as_float(x1);
                    const unsigned x2 = __float_as_uint(r1) & 0xffff0000u;
                    const unsigned x3 = __float_as_uint(r1 - __uint_as_float(x2)) & 0xffff0000u;
                    ln_cfrag[tid] = make_uint4((x1 >> 16) | x2, x1 >> 16, (x3 >> 16) | x1, x2 >> 16);
                }
            }
        }
        wait_vmcnt<0>();                               // K-tile 0 of this tile (requested under the previous tile's last K-tile) and the loads above
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (prof_on && first_tile) pt1 = prof_now();
        first_tile = false;

        const int vn = v + nwg;                        // this workgroup's next tile
        const bool has_next = vn < ntiles;
        int m0n = 0, n0n = 0;
        if (has_next) tile_origin(vn, m0n, n0n);
        const unsigned sA0 = (unsigned)m0 * (unsigned)p.lda * 2u, sW0 = (unsigned)n0 * (unsigned)p.ldw * 2u;
        const unsigned sA1 = (unsigned)m0n * (unsigned)p.lda * 2u, sW1 = (unsigned)n0n * (unsigned)p.ldw * 2u;

        f32x16 acc[F_FM][F_FN];
#pragma unroll
        for (int i = 0; i < F_FM; ++i)
#pragma unroll
            for (int j = 0; j < F_FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        frag_ab fa[2][F_FM], fb[F_FN];
        auto rd = [&](const char* row, int kk) __attribute__((always_inline)) -> frag_ab { return *(const frag_ab*)(row + (((kk * 2 + lhi) ^ fsw) << 4)); };
#pragma unroll
        for (int i = 0; i < F_FM; ++i) fa[0][i] = rd(smem + cur * F_STAGE + offA + i * 32 * 128, 0);
#pragma unroll
        for (int j = 0; j < F_FN; ++j) fb[j] = rd(smem + cur * F_STAGE + offW + j * 32 * 128, 0);
        // one k-step (16 of the K-tile's 64), column by column: 10 MFMAs on A set S; W fragment j is re-read for k-step rkk of slot rbuf right behind its last MFMA,
        // the A fragments of that k-step go to set 1 - S behind the first two; dma > 0: LDS-DMA piece q of a K-tile behind MFMA q (slot dslot, scalar offsets da / dw)
        auto kstep = [&](const int S, const int rbuf, const int rkk, const bool dma, char* dslot, unsigned da, unsigned dw) __attribute__((always_inline)) {
            const char* pa = smem + rbuf * F_STAGE + offA;
            const char* pb = smem + rbuf * F_STAGE + offW;
#pragma unroll
            for (int q = 0; q < F_FM * F_FN; ++q) {
                const int i = q % F_FM, j = q / F_FM;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[S][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q < F_FM) fa[1 - S][q] = rd(pa + q * 32 * 128, rkk);
                if (i == F_FM - 1) fb[j] = rd(pb + j * 32 * 128, rkk);
                if (dma && q < F_L) dma_piece(q, dslot, da, dw);         // (`dma` is wave-uniform: a scalar branch)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ONE code path per K-tile (three specialised copies of k-step 0 spilled 640 registers): what changes under the tile's last K-tile are scalars
        for (int kt = 0; kt < nk; ++kt) {
            const bool last = kt == nk - 1;
            char* nslot = smem + (cur ^ 1) * F_STAGE;  // released by the barrier that ended the previous K-tile
            // k-step 0 carries the LDS-DMA of the next K-tile -- of THIS tile, or K-tile 0 of the workgroup's NEXT tile under the last one
            const bool do_dma = !last || has_next;
            const unsigned da = last ? sA1 : sA0 + (unsigned)(kt + 1) * (BK * 2), dw = last ? sW1 : sW0 + (unsigned)(kt + 1) * (BK * 2);
            kstep(0, cur, 1, do_dma, nslot, da, dw);
            kstep(1, cur, 2, false, nslot, 0u, 0u);
            kstep(0, cur, 3, false, nslot, 0u, 0u);
            // every fragment of K-tile kt is in registers; K-tile kt + 1 has landed (not waited for under the last K-tile: the next tile's prologue does)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!last) wait_vmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            if (!last) cur ^= 1;
            kstep(1, cur, 0, false, nslot, 0u, 0u);    // (behind the last K-tile these reads fetch stale bytes into registers nobody uses)
        }
        if (prof_on && !has_next) pt2 = prof_now();

        // ---- fused LayerNorm, part 2: acc[m][n] += (-mean_m) * colsum_n as one more MFMA k-step; rstd_m multiplies in the epilogue
        float rs_row[F_FM];
#pragma unroll
        for (int i = 0; i < F_FM; ++i) rs_row[i] = 1.f;
        if (ln_on) {
            frag_ab la[F_FM], lb[F_FN];
#pragma unroll
            for (int i = 0; i < F_FM; ++i) {
                const uint2 h = ((const uint2*)(ln_mfrag + wr * F_TM + i * 32 + l31))[lhi];
                uint4 u = make_uint4(h.x, h.y, 0u, 0u);
                la[i] = *(frag_ab*)&u;
                rs_row[i] = ln_rs[wr * F_TM + i * 32 + l31];
            }
#pragma unroll
            for (int j = 0; j < F_FN; ++j) {
                const uint2 h = ((const uint2*)(ln_cfrag + wc * F_TN + j * 32 + l31))[lhi];
                uint4 u = make_uint4(h.x, h.y, 0u, 0u);
                lb[j] = *(frag_ab*)&u;
            }
#pragma unroll
            for (int i = 0; i < F_FM; ++i)
#pragma unroll
                for (int j = 0; j < F_FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[j], la[i], acc[i][j], 0, 0, 0);
        }
        // ---- GEGLU epilogue (gemm_kernel.h's staged form): within a 32-row W fragment accumulator groups g = 0, 1 are the value rows, g = 2, 3 the gate rows.
        // The patches live in the slot of the tile's LAST K-tile (`cur`); the other slot is receiving K-tile 0 of the next tile.
        {
            char* stg = smem + cur * F_STAGE + w * F_STG;
            bf16_t* Cb = p.C;
            auto chunk = [&](int i, int j0, auto cf_tag) __attribute__((always_inline)) {
                constexpr int CF = decltype(cf_tag)::value, OC = CF * 16, SR = OC * 2 + 16, LPR = OC / 8, RPI = 64 / LPR, NP = 32 / RPI;
                const int rr = lane / LPR, cc = (lane % LPR) * 8;
#pragma unroll
                for (int jj = 0; jj < CF; ++jj) {
                    const int j = j0 + jj;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        float o[4];
                        const float4 ba = *(const float4*)(bias_lds + wc * F_TN + j * 32 + g * 8 + lhi * 4), bg = *(const float4*)(bias_lds + wc * F_TN + j * 32 + g * 8 + lhi * 4 + 16);
                        const float bav[4] = {ba.x, ba.y, ba.z, ba.w}, bgv[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = fmaf(acc[i][j][g * 4 + r], rs_row[i], bav[r]), gt = fmaf(acc[i][j][(g + 2) * 4 + r], rs_row[i], bgv[r]);
                            o[r] = a * gelu_erf_f(gt);
                        }
                        uint2 vv; vv.x = pack_bf2(o[0], o[1]); vv.y = pack_bf2(o[2], o[3]);
                        *(uint2*)(stg + l31 * SR + (jj * 16 + g * 8 + lhi * 4) * 2) = vv;
                    }
                }
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    const int r = ps * RPI + rr, m = m0 + wr * F_TM + i * 32 + r;
                    const uint4 vv = *(const uint4*)(stg + r * SR + cc * 2);
                    const int col = (n0 + wc * F_TN) / 2 + j0 * 16 + cc;
                    *(uint4*)(Cb + (int64_t)m * p.ldc + col) = vv;
                }
            };
#pragma unroll
            for (int i = 0; i < F_FM; ++i) {
                chunk(i, 0, std::integral_constant<int, 4>{});
                chunk(i, 4, std::integral_constant<int, 1>{});
            }
        }
        if (!has_next) break;
        // every wave is through with the patches and with this tile's LayerNorm block before the next tile rewrites either
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        v = vn; m0 = m0n; n0 = n0n; cur ^= 1;
    }
    asm volatile("" :: "v"(pf_sink));
    if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
}

}  // namespace

// can tiling 24 run this launch?  (plain bf16 GEMM with the staged GEGLU epilogue, shared weights, whole tiles)
bool ff1p_eligible(const Params& p, int conv, int f8, int batch) {
    return !conv && !f8 && batch == 1 && p.n_trans_begin < 0 && p.epilogue == TMIX_EPI_GEGLU && (p.wide & 2) && !p.f8out && !p.f8copy && !p.cs_out && !p.rgb && !p.R && !p.scaleA && !p.stats_out
           && (p.M % F_BM) == 0 && (p.N % F_BN) == 0 && p.K >= 2 * BK && p.w_period == 0;
}

int launch_ff1p(Params& p, hipStream_t st) {
    constexpr int SMEM = F_RING + (F_BM + F_BN) * 16 + F_BM * 4 + F_BN * 4;
    static_assert(SMEM <= 160 * 1024, "LDS");
    static int n_cu = 0;
    if (!n_cu) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_ff1p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) TMIX_FAIL(TMIX_EARCH, "gemm: cannot query the device");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    p.tiles_m = p.M / F_BM; p.tiles_n = p.N / F_BN;
    p.group_m = 8;
    const int ntiles = p.tiles_m * p.tiles_n;
    // one workgroup per CU (a multiple of 8, so that a workgroup's tiles stay on its XCD's patch: v and v + grid are the same residue mod 8)
    int grid = ntiles < n_cu ? ntiles : (n_cu / 8) * 8;
    if (grid < 1) grid = 1;
    p.prof = tmix_prof_take(&p.prof_detail);
    tmix_prefetch_take(&p.pf, &p.pf_bytes);
    { const long long nthr = (long long)grid * F_NW * 64, lines = (p.pf_bytes + 127) >> 7;
      p.pf_per = p.pf ? (int)((lines + nthr - 1) / nthr) : 0; }
    gemm_ff1p_kernel<<<dim3(grid), F_NW * 64, SMEM, st>>>(p);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

}  // namespace tmix_gemm
