// gemm_w22.hip -- bf16 GEMM C = A * W^T on 128 x 160 tiles with a 2 x 2 wave layout (TMIX_TILE_128x160_W22 = 23).
//
// The N = 1280 / 640 projections of this path (attention out-projections, attn2 to_q, FF2; utils_lora.py:65-69,113-119 and the
// BasicTransformerBlock feed-forward behind fusion_sampling.py:340) need 160-wide tiles to put exactly one tile on each of the 256 CUs,
// and the lock-step loop of gemm_kernel.h runs them as FOUR waves of 32 x 160: every wave re-reads the whole W tile, 96 KB of fragment
// reads per 36 KB K-tile.  Here the four math waves own 64 x 80 each (4 x 5 fragments of v_mfma_f32_16x16x32_bf16): 72 KB of fragment
// reads per K-tile for the same 40 MFMA-cycles -- the LDS bytes per output of a 2 x 2 layout (HISTORY: 128 x 192 over 2 x 2 waves took the
// K-loop time of 128 x 160 for 20 % more outputs) without giving up the 160-wide tile.  Four loader waves (one per SIMD) stream the
// K-tiles by LDS-DMA exactly as tiling 21 does; ring of four stages, swizzle, tile order, weight prefetch hint and in-situ timing are
// gemm_kernel.h's.  Epilogue: the staged plain form only -- bias, folded-LayerNorm consumer, residual, bf16 store, LayerNorm row
// statistics for the next consumer, per-batch / periodic weight sets.  Everything else stays on the other tilings (gemm_conv.hip routes).
//
// MFMA roofline: 2*M*N*K flops per launch against the 2.5 PFLOP/s dense bf16 peak.
#include "gemm_kernel.h"

namespace tmix_gemm {

namespace {

constexpr int W_BM = 128, W_BN = 160, W_NS = 4, W_LW = 4, W_NW = 4;
constexpr int W_TM = 64, W_TN = 80, W_FM = 4, W_FN = 5;               // wave tile and its 16 x 16 fragments
constexpr int W_ATILE = W_BM * 128, W_BTILE = W_BN * 128, W_STAGE = W_ATILE + W_BTILE;
constexpr int W_RING = W_NS * W_STAGE;
// (LDS-DMA instructions per K-tile: 16 for A + 20 for W, dealt to the 4 -- or 3, tiling 25 -- DMA loader waves)
constexpr int W_SR = W_TN * 4 + 16;                                    // bytes per row of a wave's staging patch (80 fp32 columns + pad)
constexpr int W_PATCH = 32 * W_SR;                                     // 32-row half of a wave tile
constexpr int W_CG = W_TN / 8;                                         // 8-column groups per row of a wave tile (one 16-byte store each)
constexpr int W_PASSES = (32 * W_CG) / 64;                             // read-back passes per 32-row half: 5
static_assert((W_BM / 8) % W_LW == 0 && (W_BN / 8) % W_LW == 0 && (32 * W_CG) % 64 == 0, "geometry");
static_assert(W_NW * W_PATCH + 2 * W_BM * W_CG * 8 <= W_RING, "epilogue patches + statistics exchange fit in the staging ring");

// PFW = 1 (tiling 25): the fourth loader wave issues no LDS-DMA -- it walks W_PD K-tiles AHEAD of the ring and touches this tile's A and W lines (one dword per 128-byte
// line), so that they are in this XCD's L2 when the three DMA loaders ask for them: the ring holds 2 - 3 K-tiles (~1.5 us of lookahead), a line that comes cold through the
// fabric takes longer than that, and nothing in the LDS budget can deepen the ring.  Its own wave, because vmcnt retires in order: a slow touch in a DMA loader's queue would hold
// back the hand-over of every K-tile behind it.
constexpr int W_PD = 8;
template <int PFW>
__global__ void __launch_bounds__((W_NW + W_LW) * 64, 2) gemm_w22_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef TMIX_NO_KERNARG_TOUCH
    kernarg_touch<(int)sizeof(Params)>();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = w >= W_NW;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, (blockIdx.x | blockIdx.y) == 0, p.prof_detail);
    // the NEXT launch's weights (tmix_gemm_prefetch_next): touched by the loader waves IN FRONT of K-tile 0, whose counted vmcnt waits cover the loads.
    // (Round 5 moved the touches behind the first barrier and then spread them over the K loop -- tools/jobs5/r5g_pf.sh, r5h_pf2.sh: the prologue shrinks by what
    // the loop grows, 2.5 us per launch in front of a 9.8 MB q/k/v weight wherever they sit; the step came out 0.1 - 0.3 ms slower both times.)
    constexpr int PFU = 8;
    unsigned pf_keep[PFU];
#pragma unroll
    for (int u = 0; u < PFU; ++u) pf_keep[u] = 0;
#ifdef TMIX_W22_PF_MATH
    const bool pf_here = false;
#else
    const bool pf_here = loader;
#endif
    if (p.pf && pf_here) {
        const long long nwg = (long long)gridDim.x * gridDim.y, nth = W_LW * 64;
        const long long lines = (p.pf_bytes + 127) >> 7; const int per = p.pf_per;
        const long long first = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * nth + (tid - W_NW * 64);
#pragma unroll
        for (int u = 0; u < PFU; ++u) {
            const long long ln = first + (long long)u * nwg * nth;
            if (u < per && ln < lines) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_keep[u]) : "v"(p.pf + (ln << 7)) : "memory");
        }
    }
    // tile order: gemm_kernel.h's (each XCD owns a compact patch of group_m x (64 / group_m) tiles)
    int bid, by;
    xcd_remap_grid(bid, by);
    const int per_group = p.group_m * p.tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * p.group_m;
    const int gsize = min(p.tiles_m - first_m, p.group_m);
    const int rem = bid - grp * per_group;
    const int tile_n = rem / gsize, tile_m = first_m + (rem - tile_n * gsize);
    const int m0 = tile_m * W_BM, n0 = tile_n * W_BN;
    const int bzw = p.w_period > 0 ? (int)__umulhi((unsigned)by, p.w_magic) : by;
    const int bz = p.w_period > 0 ? (by - bzw * p.w_groups) * p.w_period + bzw : by;
    const bf16_t* Ab = p.A + (int64_t)bz * p.strideA;
    const bf16_t* Wb = p.W + (int64_t)bzw * p.strideW;
    const int nk = p.K / BK;

    if (loader) {
        // ---- loader waves: LDS-DMA issue + counted waits only.  Loader s issues instructions idx = 4 r + s of a K-tile (8 rows of 128 bytes each);
        // LDS position q of row r holds source chunk q ^ ((r >> 1) & 7) -- the permutation rides in the lane's SOURCE address
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, p.bytesA, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, p.bytesW, 0x00020000);
        const int s = w - W_NW, lrow = lane >> 3;
        constexpr int DL = W_LW - PFW, IA = W_BM / 8, L = (W_BM / 8 + W_BN / 8) / DL;        // DMA loaders; A instructions of a K-tile; instructions per loader
        static_assert((W_BM / 8 + W_BN / 8) % DL == 0 && (W_NS - 2) * L <= 63, "loader geometry");
        if (PFW && s == DL) {
            // ---- L2 prefetcher wave: line slot i * 64 + lane = A row (< 128) or W row of the tile
            constexpr int NLI = (W_BM + W_BN + 63) / 64;
            const char* lp[NLI];
#pragma unroll
            for (int i = 0; i < NLI; ++i) {
                const int line = i * 64 + lane;
                lp[i] = line < W_BM ? (const char*)Ab + (int64_t)min(m0 + line, p.M - 1) * p.lda * 2
                      : line < W_BM + W_BN ? (const char*)Wb + (int64_t)min(n0 + line - W_BM, p.N - 1) * p.ldw * 2 : nullptr;
            }
            unsigned sink = 0;
            auto touch = [&](int kp) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < NLI; ++i)
                    if (lp[i]) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(lp[i] + (int64_t)kp * (BK * 2)) : "memory");
            };
            for (int kp = W_NS - 1; kp <= W_PD && kp < nk; ++kp) touch(kp);
            __builtin_amdgcn_s_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + W_PD + 1 < nk) touch(kt + W_PD + 1);
                __builtin_amdgcn_s_barrier();
            }
            wait_vmcnt<0>();
            asm volatile("" :: "v"(sink));
#pragma unroll
            for (int u = 0; u < PFU; ++u) asm volatile("" :: "v"(pf_keep[u]));
            if (p.stats_out) __syncthreads();
            return;
        }
        // DMA loaders: instruction g = DL r + s of a K-tile: g < 16 -> A rows 8 g .., else W rows 8 (g - 16) ..; the swizzle depends on the instruction's parity only
        const unsigned sw0 = (unsigned)(((lane & 7) ^ ((lane >> 4) & 7)) * 16), sw1 = (unsigned)(((lane & 7) ^ ((4 + (lane >> 4)) & 7)) * 16);
        const unsigned aoff0 = (unsigned)(m0 + lrow) * (unsigned)p.lda * 2u + sw0, dA = 16u * (unsigned)p.lda + sw1 - sw0;
        const unsigned woff0 = (unsigned)(n0 + lrow) * (unsigned)p.ldw * 2u + sw0, dW = 16u * (unsigned)p.ldw + sw1 - sw0;
        const unsigned amax0 = (unsigned)(p.M - 1) * (unsigned)p.lda * 2u + sw0, wmax0 = (unsigned)(p.N - 1) * (unsigned)p.ldw * 2u + sw0, dS = sw1 - sw0;
        auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
            char* sA = smem + buf * W_STAGE;
            char* sW = sA + W_ATILE;
#pragma unroll
            for (int r = 0; r < L; ++r) {
                const int g = r * DL + s;              // wave-uniform
                if (g < IA) {
                    const unsigned odd = 0u - (unsigned)(g & 1);
                    blds16(rsA, min(aoff0 + (dA & odd) + (unsigned)(g >> 1) * (unsigned)(32 * p.lda), amax0 + (dS & odd)), (unsigned)kt * (BK * 2), sA + g * 1024);
                } else {
                    const int idx = g - IA; const unsigned odd = 0u - (unsigned)(idx & 1);
                    blds16(rsW, min(woff0 + (dW & odd) + (unsigned)(idx >> 1) * (unsigned)(32 * p.ldw), wmax0 + (dS & odd)), (unsigned)kt * (BK * 2), sW + idx * 1024);
                }
            }
        };
        constexpr int PRE = 2;
#pragma unroll
        for (int t = 0; t < PRE; ++t)
            if (t < nk) stage(t, t);
        if (nk >= PRE) wait_vmcnt<(PRE - 1) * L>(); else wait_vmcnt<0>();
#pragma unroll
        for (int u = 0; u < PFU; ++u) asm volatile("" :: "v"(pf_keep[u]));
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = PRE; t < W_NS - 1; ++t)
            if (t < nk) stage(t, t);
        int nxt = W_NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + W_NS - 1 < nk;
            if (more) stage(nxt, kt + W_NS - 1);      // its ring slot was released by the barrier that ended iteration kt - 1
            if (more) wait_vmcnt<(W_NS - 2) * L>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            nxt = (nxt + 1 == W_NS) ? 0 : nxt + 1;
        }
        if (p.stats_out) __syncthreads();             // the math waves' statistics exchange has one more barrier
        return;
    }

    // ---------------------------------------------------------------- math waves
    const int wr = w >> 1, wc = w & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    // fused LayerNorm (consumer side), part 1 + the tile's bias: as in gemm_kernel.h (thread t owns tile row t and tile column t)
    uint4* ln_mfrag = (uint4*)(smem + W_RING);
    uint4* ln_cfrag = ln_mfrag + W_BM;
    float* ln_rs = (float*)(ln_cfrag + W_BN);
    float* bias_lds = ln_rs + W_BM;
    float bias_r = 0.f;
    if (p.bias && tid < W_BN && n0 + tid < p.N) bias_r = (p.bias + (int64_t)bzw * p.strideBias)[n0 + tid];
    constexpr int PU = 16;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    u32x2 lnv[PU];
    float ln_cs = 0.f;
    const bool ln_on = p.ln_stats != nullptr;
    if (ln_on) {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_stats + (int64_t)bz * p.strideLnStats), 0,
                                                                              (int)(p.ln_parts * p.ldLnStats * 8), 0x00020000);
        if (tid < W_BM) {
            const int lnm = min(m0 + tid, p.M - 1);
#pragma unroll
            for (int q = 0; q < PU; ++q)
                if (q < p.ln_parts) lnv[q] = __builtin_amdgcn_raw_buffer_load_b64(rsS, lnm * 8, q * (int)p.ldLnStats * 8, 0);
        }
        if (tid < W_BN) ln_cs = (p.ln_colsum + (int64_t)bzw * p.strideLnColsum)[min(n0 + tid, p.N - 1)];
    }
    // (the operand pieces of the rank-1 term -mean_m * colsum_n: x = x1 + x2 + x3 in bf16 pieces, see gemm_kernel.h ln_reduce)
    if (tid < W_BN) bias_lds[tid] = bias_r;
    if (ln_on) {
        if (tid < W_BM) {
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
        if (tid < W_BN) {
            const unsigned x1 = __float_as_uint(ln_cs) & 0xffff0000u;
            const float r1 = ln_cs - __uint_as_float(x1);
            const unsigned x2 = __float_as_uint(r1) & 0xffff0000u;
            const unsigned x3 = __float_as_uint(r1 - __uint_as_float(x2)) & 0xffff0000u;
            ln_cfrag[tid] = make_uint4((x1 >> 16) | x2, x1 >> 16, (x3 >> 16) | x1, x2 >> 16);
        }
    }

    f32x4 acc[W_FM][W_FN];
#pragma unroll
    for (int i = 0; i < W_FM; ++i)
#pragma unroll
        for (int j = 0; j < W_FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment (i, k-step kk) of a stage: row wr * 64 + i * 16 + l15, source chunk 4 kk + lg at position chunk ^ ((row >> 1) & 7)
    const int fsw = (lane >> 1) & 7;
    const int offA = (wr * W_TM + l15) * 128, offW = W_ATILE + (wc * W_TN + l15) * 128;
    const int c0 = ((0 + lg) ^ fsw) << 4, c1 = ((4 + lg) ^ fsw) << 4;
    // residual rows in the read-back layout of the epilogue (slot = pass * 64 + lane -> row slot / 10 of the 32-row half, columns (slot % 10) * 8 ..):
    // requested in front of the last K-tile's MFMAs so that their latency rides under compute
    const bf16_t* Rb = p.R ? p.R + (int64_t)bz * p.strideR : nullptr;
    uint4 rw[2 * W_PASSES];
    auto epi_prefetch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ps = 0; ps < W_PASSES; ++ps) {
                const int slot = ps * 64 + lane, row = slot / W_CG, cg = slot - row * W_CG;
                const int m = m0 + wr * W_TM + h * 32 + row, nc = n0 + wc * W_TN + cg * 8;
                rw[h * W_PASSES + ps] = make_uint4(0u, 0u, 0u, 0u);
                if (Rb && m < p.M) rw[h * W_PASSES + ps] = *(const uint4*)(Rb + (int64_t)m * p.ldr + nc);
            }
    };

#ifdef TMIX_W22_PF_MATH
    // (dev variant) the touches ride in the MATH waves' queue, which nothing waits on until the residual rows are asked for: the loaders' first K-tile is not behind them
    unsigned pf_sink = 0;
    if (p.pf) {
        static_assert(W_NW == W_LW, "the host deals the lines out over W_LW * 64 threads per workgroup");
        const long long nwg = (long long)gridDim.x * gridDim.y, nth = W_NW * 64;
        const long long lines = (p.pf_bytes + 127) >> 7; const int per = p.pf_per;
        const long long first = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * nth + tid;
#pragma unroll
        for (int u = 0; u < PFU; ++u) {
            const long long ln = first + (long long)u * nwg * nth;
            if (u < per && ln < lines) asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(p.pf + (ln << 7)) : "memory");
        }
    }
#endif
    __builtin_amdgcn_s_barrier();                      // K-tile 0 has landed
    asm volatile("" ::: "memory");
    if (prof_on) pt1 = prof_now();
    frag_ab fa[2][W_FM], fb[2][W_FN];
#pragma unroll
    for (int i = 0; i < W_FM; ++i) fa[0][i] = *(const frag_ab*)(smem + offA + i * 16 * 128 + c0);
#pragma unroll
    for (int j = 0; j < W_FN; ++j) fb[0][j] = *(const frag_ab*)(smem + offW + j * 16 * 128 + c0);
    // one k-step (32 of the K-tile's 64): 20 MFMAs on register set S with the 9 fragment reads of the NEXT k-step (stage rbuf, chunk offset rc,
    // into set 1 - S) behind the first nine of them; issue order pinned
    auto kstep = [&](const int S, const int rbuf, const int rc) __attribute__((always_inline)) {
        const char* pa = smem + rbuf * W_STAGE + offA + rc;
        const char* pb = smem + rbuf * W_STAGE + offW + rc;
#pragma unroll
        for (int q = 0; q < W_FM * W_FN; ++q) {
            const int i = q / W_FN, j = q - i * W_FN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[S][j], fa[S][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < W_FM) fa[1 - S][q] = *(const frag_ab*)(pa + q * 16 * 128);
            else if (q < W_FM + W_FN) fb[1 - S][q - W_FM] = *(const frag_ab*)(pb + (q - W_FM) * 16 * 128);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
#ifdef TMIX_W22_PF_MATH
        if (kt == nk - 1) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
        if (kt == nk - 1) { epi_prefetch(); __builtin_amdgcn_sched_barrier(0); }
        kstep(0, cur, c1);
        // every fragment of tile kt is in registers (its ring slot may be restaged after the barrier); tile kt + 1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        cur = (cur + 1 == W_NS) ? 0 : cur + 1;
        kstep(1, cur, c0);                             // (behind the last K-tile these reads fetch a stale slot into registers nobody uses)
    }
    if (prof_on) pt2 = prof_now();

    // ---- fused LayerNorm (consumer side), part 2: Linear(LN(x))[m][n] = rstd_m * (acc[m][n] - mean_m * colsum_n) + t_n -- the rank-1 term is one
    // more MFMA k-step per fragment (lane groups 0 / 1 carry the two 4-value halves of the operand pieces, groups 2 / 3 zeros)
    if (ln_on) {
        frag_ab la[W_FM], lb[W_FN];
#pragma unroll
        for (int i = 0; i < W_FM; ++i) {
            uint2 h = make_uint2(0u, 0u);
            if (lg < 2) h = ((const uint2*)(ln_mfrag + wr * W_TM + i * 16 + l15))[lg];
            uint4 u = make_uint4(h.x, h.y, 0u, 0u);
            la[i] = *(frag_ab*)&u;
        }
#pragma unroll
        for (int j = 0; j < W_FN; ++j) {
            uint2 h = make_uint2(0u, 0u);
            if (lg < 2) h = ((const uint2*)(ln_cfrag + wc * W_TN + j * 16 + l15))[lg];
            uint4 u = make_uint4(h.x, h.y, 0u, 0u);
            lb[j] = *(frag_ab*)&u;
        }
#pragma unroll
        for (int i = 0; i < W_FM; ++i)
#pragma unroll
            for (int j = 0; j < W_FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lb[j], la[i], acc[i][j], 0, 0, 0);
    }

    // ---- staged epilogue.  16 x 16 accumulator: lane holds row (of A) l15, columns 4 lg .. 4 lg + 3 of fragment (i, j).  Per 32-row half the wave
    // parks its 32 x 80 fp32 block in a private LDS patch and reads it back row-major: 10 lanes x 8 columns per row, 16-byte residual
    // loads (prefetched) and C stores.  Per element: fma(acc, rstd, bias) + residual, rounded to bf16 -- the operations of the other tilings.
    bf16_t* Cb = p.C + (int64_t)bz * p.strideC;
    char* stg = smem + w * W_PATCH;
    float2* part = (float2*)(smem + W_NW * W_PATCH);         // [2 wave columns][128 rows][10 column groups] row-statistics partials
    float2* sto = p.stats_out ? (float2*)(p.stats_out + (int64_t)bz * p.strideStatsOut) : nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < W_FN; ++j)
                *(f32x4*)(stg + (ii * 16 + l15) * W_SR + (j * 16 + 4 * lg) * 4) = acc[2 * h + ii][j];
#pragma unroll
        for (int ps = 0; ps < W_PASSES; ++ps) {
            const int slot = ps * 64 + lane, row = slot / W_CG, cg = slot - row * W_CG;
            const int ml = wr * W_TM + h * 32 + row, m = m0 + ml, nl = wc * W_TN + cg * 8;
            const float4 v0 = *(const float4*)(stg + row * W_SR + cg * 32), v1 = *(const float4*)(stg + row * W_SR + cg * 32 + 16);
            const float4 b0 = *(const float4*)(bias_lds + nl), b1 = *(const float4*)(bias_lds + nl + 4);
            const float rs = ln_on ? ln_rs[ml] : 1.f;
            float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const float bq[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fmaf(o[k], rs, bq[k]);
            const uint4 rq = rw[h * W_PASSES + ps];
            const unsigned ru[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[2 * k] += __uint_as_float(ru[k] << 16); o[2 * k + 1] += __uint_as_float(ru[k] & 0xffff0000u); }
            uint4 v;
            v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]); v.z = pack_bf2(o[4], o[5]); v.w = pack_bf2(o[6], o[7]);
            const bool ok = m < p.M;
            if (ok) *(uint4*)(Cb + (int64_t)m * p.ldc + n0 + nl) = v;
            if (sto) {                                 // statistics of the values AS STORED
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(u[k] << 16), hi = __uint_as_float(u[k] & 0xffff0000u);
                    a1 += lo + hi; a2 = fmaf(lo, lo, a2); a2 = fmaf(hi, hi, a2);
                }
                part[(wc * W_BM + ml) * W_CG + cg] = make_float2(a1, a2);
            }
        }
    }
    if (sto) {
        // the 20 partials of a row (2 wave columns x 10 column groups) are added in a fixed order by the row's thread; one {sum, sum of squares}
        // per (column tile, row) goes to stats_out[tile_n][m] -- the consumer adds the tiles_n partials (gemm_kernel.h)
        __syncthreads();
        if (tid < W_BM && m0 + tid < p.M) {
            float2 t = make_float2(0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < W_CG; ++g) { const float2 u = part[(c * W_BM + tid) * W_CG + g]; t.x += u.x; t.y += u.y; }
            sto[(int64_t)tile_n * p.ldStatsOut + m0 + tid] = t;
        }
    }
    if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
}

}  // namespace

// can tiling 23 run this launch?  (plain GEMM, bf16, staged plain epilogue without activation / row-group bias / transposed region / e4m3 copy / column statistics)
bool w22_eligible(const Params& p, int conv, int f8) {
    return !conv && !f8 && p.n_trans_begin < 0 && p.epilogue == TMIX_EPI_NONE && (p.wide & 1) && !p.f8copy && !p.cs_out && !p.rgb && !p.scaleA
           && (p.N % W_BN) == 0 && (!p.R || (p.ldr % 8) == 0);
}

template <int PFW> static int launch_w22_t(Params& p, int batch, hipStream_t st) {
    constexpr int SMEM = W_RING + (W_BM + W_BN) * 16 + W_BM * 4 + W_BN * 4;
    static_assert(SMEM <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_w22_kernel<PFW>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    p.tiles_m = (p.M + W_BM - 1) / W_BM; p.tiles_n = (p.N + W_BN - 1) / W_BN;
    p.group_m = 8;
    // the kernels remap the LINEAR workgroup id over the whole (tiles, slices) grid in 32-bit arithmetic (common.h xcd_remap_grid)
    if ((int64_t)p.tiles_m * p.tiles_n * batch > 0x7fffffffLL) TMIX_FAIL(TMIX_ESHAPE, "gemm: %lld x %d workgroups exceed the 32-bit linear grid id", (long long)p.tiles_m * p.tiles_n, batch);
    dim3 grid(p.tiles_m * p.tiles_n, batch, 1);
    p.prof = tmix_prof_take(&p.prof_detail);
    tmix_prefetch_take(&p.pf, &p.pf_bytes);
    { const long long nthr = (long long)grid.x * grid.y * W_LW * 64, lines = (p.pf_bytes + 127) >> 7;
      p.pf_per = p.pf ? (int)((lines + nthr - 1) / nthr) : 0; }
    gemm_w22_kernel<PFW><<<grid, (W_NW + W_LW) * 64, SMEM, st>>>(p);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

#ifdef TMIX_EXPERIMENTAL_TILINGS
int launch_w22(Params& p, int batch, hipStream_t st, int l2_prefetcher) { return l2_prefetcher ? launch_w22_t<1>(p, batch, st) : launch_w22_t<0>(p, batch, st); }
#else       // the prefetcher-wave form (tiling 25: 50 % slower) is compiled into dev variants only
int launch_w22(Params& p, int batch, hipStream_t st, int) { return launch_w22_t<0>(p, batch, st); }
#endif

}  // namespace tmix_gemm
