// gemm_qattn.hip -- attn2.to_q and the cross-attention behind it as ONE launch (tmix_gemm_q_cross_attn).
//
// Replaces the pair tmix_gemm_bf16 (q = LN(h) Wq^T) -> tmix_attn_fwd (softmax(q K^T / sqrt(d)) V against the <= 80 cached prompt keys) of every
// BasicTransformerBlock's attn2 -- the patched forward of utils_custom.py:56-106 / utils_lora.py:65-69,101-111 -- 70 times per UNet call.  In the
// captured step these launches are bound by bytes through the fabric and by what surrounds a launch, not by their loops: the q
// tensor made a 21 MB round trip through HBM-side memory per layer and the 77-key attention kernel was all launch cost (7.5 us at 4.7 % MFMA-busy).
//
// Tile: 64 query rows x 320 output columns = FIVE heads; five math waves (wave h owns head h: a 64 x 64 accumulator of 4 x 4 fragments of
// v_mfma_f32_16x16x32_bf16) + three loader waves streaming 48 KB K-tiles by LDS-DMA through a 3-deep ring (gemm_kernel.h's swizzle, counted vmcnt).
// M = 4096, N = 1280 is 256 tiles -- one per CU.  Five math waves sit on four SIMDs (one SIMD carries two): the K loop of these launches waits for
// operands ~1250 cycles per K-tile in situ, the doubly-loaded SIMD needs 1024 for its MFMAs.
// Behind the K loop a wave holds q^T of its head in the accumulator layout (lane: query l15 of fragment i, d = 16 j + 4 lg + r) -- which IS the
// B-operand layout of S^T = K q^T once the contraction index d is permuted the same way on the K side (two 8-byte loads per K fragment).  From
// there on it is attn_small_kernel's arithmetic in registers: exact maximum per query, exp2, P^T feeds O^T = V^T P^T as the next B operand (key
// slots permuted as in attention.hip), row sums on the matrix core.  K fragments are requested from L2 under the last K-tiles; V^T of the five
// heads is brought into the (then free) ring by the loaders while the math waves compute the scores.  O^T comes out in the accumulator layout again
// and leaves through a staged 16-byte-per-lane store.
//
// MFMA roofline: 2*M*N*K (projection) + 4*M*Skv*N (attention) flops per launch against the 2.5 PFLOP/s dense bf16 peak.
#include "gemm_kernel.h"

namespace tmix_gemm {

struct QAParams {
    Params g;
    const bf16_t* Kc; int64_t ldk, strideK;        // cached keys   [images][Skv][ldk]  (column = head * 64 + d)
    const bf16_t* Vt; int64_t ldvt, strideVt;      // cached V^T    [images][N][ldvt]   (row = head * 64 + d)
    bf16_t* O; int64_t ldo;                        // attention output [batch * M][ldo]
    int rows_per_image, Skv; float scale_log2e;
};

namespace {

#ifdef TMIX_QATTN_H4      // dev A/B builds (VERDICT r5 item 6 ii): FOUR heads per tile -- 64 x 256, four math + four loader waves, 40 KB K-tiles, 320 tiles at M = 4096 (1.25 rounds)
constexpr int Q_BM = 64, Q_BN = 256, Q_NS = 3, Q_NW = 4, Q_LW = 4;
#else
constexpr int Q_BM = 64, Q_BN = 320, Q_NS = 3, Q_NW = 5, Q_LW = 3;
#endif
constexpr int Q_ATILE = Q_BM * 128, Q_BTILE = Q_BN * 128, Q_STAGE = Q_ATILE + Q_BTILE, Q_RING = Q_NS * Q_STAGE;
constexpr int Q_IA = Q_BM / 8, Q_IB = Q_BN / 8, Q_L = (Q_IA + Q_IB) / Q_LW;         // 8 + 40 LDS-DMA instructions per K-tile, 16 per loader
static_assert((Q_IA + Q_IB) % Q_LW == 0 && (Q_NS - 2) * Q_L <= 63, "loader geometry");
constexpr int Q_LDV = 80;                           // keys per V^T row (Skv <= 80 rounded up to 8: the KV cache's leading dimension)
constexpr int Q_VH = 64 * Q_LDV * 2;                // bytes of one head's V^T: 64 rows of 160 bytes, contiguous
constexpr int Q_VBYTES = Q_NW * Q_VH;               // 51,200 bytes at the front of the ring (one V^T per head of the tile)
constexpr int Q_SR = 64 * 4 + 16;                   // bytes per row of a wave's output patch (64 fp32 columns + pad)
constexpr int Q_PATCH = 32 * Q_SR;
constexpr int Q_POFF = (Q_VBYTES + 1023) / 1024 * 1024;
static_assert(Q_POFF + Q_NW * Q_PATCH <= Q_RING, "V^T + output patches fit in the staging ring");

__global__ void __launch_bounds__((Q_NW + Q_LW) * 64, 2) gemm_qattn_kernel(const QAParams qp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Params p = qp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = w >= Q_NW;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, (blockIdx.x | blockIdx.y) == 0, p.prof_detail);
    // the NEXT launch's weights (tmix_gemm_prefetch_next): touched by the loader waves in front of K-tile 0 (gemm_kernel.h)
    constexpr int PFU = 8;
    unsigned pf_keep[PFU];
#pragma unroll
    for (int u = 0; u < PFU; ++u) pf_keep[u] = 0;
    if (p.pf && loader) {
        const long long nwg = (long long)gridDim.x * gridDim.y, nth = Q_LW * 64;
        const long long lines = (p.pf_bytes + 127) >> 7; const int per = p.pf_per;
        const long long first = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * nth + (tid - Q_NW * 64);
#pragma unroll
        for (int u = 0; u < PFU; ++u) {
            const long long ln = first + (long long)u * nwg * nth;
            if (u < per && ln < lines) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_keep[u]) : "v"(p.pf + (ln << 7)) : "memory");
        }
    }
    int bid, by;
    xcd_remap_grid(bid, by);
    const int per_group = p.group_m * p.tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * p.group_m;
    const int gsize = min(p.tiles_m - first_m, p.group_m);
    const int rem = bid - grp * per_group;
    const int tile_n = rem / gsize, tile_m = first_m + (rem - tile_n * gsize);
    const int m0 = tile_m * Q_BM, n0 = tile_n * Q_BN;
    const int bzw = p.w_period > 0 ? (int)__umulhi((unsigned)by, p.w_magic) : by;
    const int bz = p.w_period > 0 ? (by - bzw * p.w_groups) * p.w_period + bzw : by;
    const bf16_t* Ab = p.A + (int64_t)bz * p.strideA;
    const bf16_t* Wb = p.W + (int64_t)bzw * p.strideW;
    const int nk = p.K / BK;
    // the image whose prompt keys this tile attends to (a tile never straddles two images: rows_per_image % 64 == 0)
    const int img = bz * (p.M / qp.rows_per_image) + m0 / qp.rows_per_image;

    if (loader) {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, p.bytesA, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, p.bytesW, 0x00020000);
        const int s = w - Q_NW, lrow = lane >> 3;
        // instruction g = 3 r + s of a K-tile: g < 8 -> A rows 8 g .., else W rows 8 (g - 8) ..; LDS position q of row r holds source chunk q ^ ((r >> 1) & 7),
        // which depends on the instruction's parity only
        // (named scalars, not arrays: an array indexed by the wave-uniform parity went to scratch memory, whose loads share the DMA's vmcnt)
        const unsigned sw0 = (unsigned)(((lane & 7) ^ ((lane >> 4) & 7)) * 16), sw1 = (unsigned)(((lane & 7) ^ ((4 + (lane >> 4)) & 7)) * 16);
        const unsigned a_row = (unsigned)(m0 + lrow) * (unsigned)p.lda * 2u, w_row = (unsigned)(n0 + lrow) * (unsigned)p.ldw * 2u;
        const unsigned aoff0 = a_row + sw0, dA = 16u * (unsigned)p.lda + sw1 - sw0;         // odd instructions: + dA (mask arithmetic below: a select between two
        const unsigned woff0 = w_row + sw0, dW = 16u * (unsigned)p.ldw + sw1 - sw0;         // by-reference captures became a select between two stack slots)
        const unsigned amax0 = (unsigned)(p.M - 1) * (unsigned)p.lda * 2u + sw0, wmax0 = (unsigned)(p.N - 1) * (unsigned)p.ldw * 2u + sw0, dS = sw1 - sw0;
        auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
            char* sA = smem + buf * Q_STAGE;
            char* sW = sA + Q_ATILE;
#pragma unroll
            for (int r = 0; r < Q_L; ++r) {
                const int g = r * Q_LW + s;            // wave-uniform
                if (g < Q_IA) {
                    const unsigned odd = 0u - (unsigned)(g & 1);
                    blds16(rsA, min(aoff0 + (dA & odd) + (unsigned)(g >> 1) * (unsigned)(32 * p.lda), amax0 + (dS & odd)), (unsigned)kt * (BK * 2), sA + g * 1024);
                } else {
                    const int idx = g - Q_IA; const unsigned odd = 0u - (unsigned)(idx & 1);
                    blds16(rsW, min(woff0 + (dW & odd) + (unsigned)(idx >> 1) * (unsigned)(32 * p.ldw), wmax0 + (dS & odd)), (unsigned)kt * (BK * 2), sW + idx * 1024);
                }
            }
        };
        constexpr int PRE = Q_NS - 1;
#pragma unroll
        for (int t = 0; t < PRE; ++t)
            if (t < nk) stage(t, t);
        if (nk >= PRE) wait_vmcnt<(PRE - 1) * Q_L>(); else wait_vmcnt<0>();
#pragma unroll
        for (int u = 0; u < PFU; ++u) asm volatile("" :: "v"(pf_keep[u]));
        __builtin_amdgcn_s_barrier();
        int nxt = Q_NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + Q_NS - 1 < nk;
            if (more) stage(nxt, kt + Q_NS - 1);
            if (more) wait_vmcnt<(Q_NS - 2) * Q_L>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            nxt = (nxt + 1 == Q_NS) ? 0 : nxt + 1;
        }
        // the ring is free (every math wave has its last fragments in registers): V^T of the tile's five heads -- 5 x 10 KB, contiguous -- goes to its front
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(qp.Vt + (int64_t)img * qp.strideVt + (int64_t)n0 * Q_LDV), 0, Q_VBYTES, 0x00020000);
        for (int q = s; q < Q_VBYTES / 1024; q += Q_LW) blds16(rsV, (unsigned)(q * 1024 + lane * 16), 0u, smem + q * 1024);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                  // V^T is visible to the math waves
        return;
    }

    // ---------------------------------------------------------------- math waves: wave h = head h of the tile
    const int h = w;
    const int l15 = lane & 15, lg = lane >> 4;
    uint4* ln_mfrag = (uint4*)(smem + Q_RING);
    uint4* ln_cfrag = ln_mfrag + Q_BM;
    float* ln_rs = (float*)(ln_cfrag + Q_BN);
    float* bias_lds = ln_rs + Q_BM;
    float bias_r = 0.f;
    if (p.bias && tid < Q_BN && n0 + tid < p.N) bias_r = (p.bias + (int64_t)bzw * p.strideBias)[n0 + tid];
    constexpr int PU = 16;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    u32x2 lnv[PU];
    float ln_cs = 0.f;
    const bool ln_on = p.ln_stats != nullptr;
    if (ln_on) {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_stats + (int64_t)bz * p.strideLnStats), 0,
                                                                              (int)(p.ln_parts * p.ldLnStats * 8), 0x00020000);
        if (tid < Q_BM) {
            const int lnm = min(m0 + tid, p.M - 1);
#pragma unroll
            for (int q = 0; q < PU; ++q)
                if (q < p.ln_parts) lnv[q] = __builtin_amdgcn_raw_buffer_load_b64(rsS, lnm * 8, q * (int)p.ldLnStats * 8, 0);
        }
        if (tid < Q_BN) ln_cs = (p.ln_colsum + (int64_t)bzw * p.strideLnColsum)[min(n0 + tid, p.N - 1)];
    }
    if (tid < Q_BN) bias_lds[tid] = bias_r;
    if (ln_on) {                                       // (gemm_kernel.h ln_reduce: rstd per row, and the bf16 operand pieces of -mean_m * colsum_n)
        if (tid < Q_BM) {
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
        if (tid < Q_BN) {
            const unsigned x1 = __float_as_uint(ln_cs) & 0xffff0000u;
            const float r1 = ln_cs - __uint_as_float(x1);
            const unsigned x2 = __float_as_uint(r1) & 0xffff0000u;
            const unsigned x3 = __float_as_uint(r1 - __uint_as_float(x2)) & 0xffff0000u;
            ln_cfrag[tid] = make_uint4((x1 >> 16) | x2, x1 >> 16, (x3 >> 16) | x1, x2 >> 16);
        }
    }

    f32x4 acc[4][4];                                   // [query fragment i][d fragment j]: lane holds query 16 i + l15, d = 16 j + 4 lg + r
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fsw = (lane >> 1) & 7;
    const int offA = l15 * 128, offW = Q_ATILE + (h * 64 + l15) * 128;
    const int c0 = ((0 + lg) ^ fsw) << 4, c1 = ((4 + lg) ^ fsw) << 4;

    // K fragments of this head (A operand of S^T = K q^T): row = key slot (f, l15) -> key 32 (f >> 1) + 8 (l15 >> 2) + 4 (f & 1) + (l15 & 3) (attention.hip's
    // permutation: P^T then feeds the PV MFMA from registers), contraction slots of lane group lg in k-step s = d 32 s + 4 lg .. + 3 and 32 s + 16 + 4 lg .. + 3
    // (what the accumulator layout gives q^T).  Requested under the last K-tiles.
    const bf16_t* Kh = qp.Kc + (int64_t)img * qp.strideK + n0 + h * 64;
    uint2 kq[6][2][2];
    auto k_prefetch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            int key = 32 * (f >> 1) + 8 * (l15 >> 2) + 4 * (f & 1) + (l15 & 3); if (key > qp.Skv - 1) key = qp.Skv - 1;
            const bf16_t* row = Kh + (int64_t)key * qp.ldk + 4 * lg;
#pragma unroll
            for (int s = 0; s < 2; ++s) { kq[f][s][0] = *(const uint2*)(row + 32 * s); kq[f][s][1] = *(const uint2*)(row + 32 * s + 16); }
        }
    };

    __builtin_amdgcn_s_barrier();                      // K-tile 0 has landed
    asm volatile("" ::: "memory");
    if (prof_on) pt1 = prof_now();
    frag_ab fa[2][4], fb[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = *(const frag_ab*)(smem + offA + i * 16 * 128 + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = *(const frag_ab*)(smem + offW + j * 16 * 128 + c0);
    auto kstep = [&](const int S, const int rbuf, const int rc) __attribute__((always_inline)) {
        const char* pa = smem + rbuf * Q_STAGE + offA + rc;
        const char* pb = smem + rbuf * Q_STAGE + offW + rc;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = q >> 2, j = q & 3;
#ifdef TMIX_QATTN_ABL4      // dev ablation (tools/build_variant.sh): the fifth math wave (the second one on its SIMD) issues no MFMAs in the K loop -- wrong results, the loop's
            if (h < 4)               // time without the doubly-loaded SIMD
#endif
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[S][j], fa[S][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 4) fa[1 - S][q] = *(const frag_ab*)(pa + q * 16 * 128);
            else if (q < 8) fb[1 - S][q - 4] = *(const frag_ab*)(pb + (q - 4) * 16 * 128);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int cur = 0;
    const int kpre = nk >= 3 ? nk - 3 : 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt == kpre) { k_prefetch(); __builtin_amdgcn_sched_barrier(0); }
        kstep(0, cur, c1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        cur = (cur + 1 == Q_NS) ? 0 : cur + 1;
        kstep(1, cur, c0);
    }
    if (prof_on) pt2 = prof_now();

    // ---- fused LayerNorm (consumer side), part 2: one more MFMA k-step adds -mean_m * colsum_n; rstd_m multiplies with the bias below
    if (ln_on) {
        frag_ab la[4], lb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 v = make_uint2(0u, 0u);
            if (lg < 2) v = ((const uint2*)(ln_mfrag + i * 16 + l15))[lg];
            uint4 u = make_uint4(v.x, v.y, 0u, 0u);
            la[i] = *(frag_ab*)&u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 v = make_uint2(0u, 0u);
            if (lg < 2) v = ((const uint2*)(ln_cfrag + h * 64 + j * 16 + l15))[lg];
            uint4 u = make_uint4(v.x, v.y, 0u, 0u);
            lb[j] = *(frag_ab*)&u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lb[j], la[i], acc[i][j], 0, 0, 0);
    }

    // ---- q^T as the B operand of S^T: q = (acc * rstd + bias) * scale * log2(e), rounded to bf16; k-step s takes d fragments 2 s and 2 s + 1
    uint4 qb[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float rs = ln_on ? ln_rs[i * 16 + l15] : 1.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float4 b4 = *(const float4*)(bias_lds + h * 64 + (2 * s + jj) * 16 + 4 * lg);
                const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[jj * 4 + r] = fmaf(acc[i][2 * s + jj][r], rs, bq[r]) * qp.scale_log2e;
            }
            qb[i][s] = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
        }
    }
    frag_ab kf[6][2];
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
        for (int s = 0; s < 2; ++s) { uint4 u = make_uint4(kq[f][s][0].x, kq[f][s][0].y, kq[f][s][1].x, kq[f][s][1].y); kf[f][s] = *(frag_ab*)&u; }
    frag_ab ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const char* vbase = smem + h * Q_VH + l15 * (Q_LDV * 2) + lg * 16;     // V^T row d = 16 jd + l15 at + jd * 16 * 160, keys 32 t + 8 lg .. at + t * 64
    char* stg = smem + Q_POFF + w * Q_PATCH;
    bf16_t* Ob = qp.O + ((int64_t)bz * p.M) * qp.ldo + n0 + h * 64;
    bool v_ready = false;
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {                   // 32 queries at a time (query fragments 2 ih, 2 ih + 1)
        f32x4 sc[6][2];
#pragma unroll
        for (int f = 0; f < 6; ++f)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                frag_ab q0, q1;
                __builtin_memcpy(&q0, &qb[2 * ih + ii][0], 16);
                __builtin_memcpy(&q1, &qb[2 * ih + ii][1], 16);
                sc[f][ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][0], q0, zero, 0, 0, 0);
                sc[f][ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][1], q1, sc[f][ii], 0, 0, 0);
            }
        uint32_t pb[2][3][4];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            float m = -INFINITY;
#pragma unroll
            for (int f = 0; f < 6; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (32 * (f >> 1) + 8 * lg + 4 * (f & 1) + r >= qp.Skv) sc[f][ii][r] = -INFINITY;
                    m = fmaxf(m, sc[f][ii][r]);
                }
            {   // over the four lane groups of this query column
                auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                m = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
                auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                m = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
            }
#pragma unroll
            for (int f = 0; f < 6; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[f][ii][r] = __builtin_amdgcn_exp2f(sc[f][ii][r] - m);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                pb[ii][t][0] = pack_bf2(sc[2 * t][ii][0], sc[2 * t][ii][1]);
                pb[ii][t][1] = pack_bf2(sc[2 * t][ii][2], sc[2 * t][ii][3]);
                pb[ii][t][2] = pack_bf2(sc[2 * t + 1][ii][0], sc[2 * t + 1][ii][1]);
                pb[ii][t][3] = pack_bf2(sc[2 * t + 1][ii][2], sc[2 * t + 1][ii][3]);
            }
        }
        if (!v_ready) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); v_ready = true; }      // V^T has landed (loader waves)
        f32x4 o[4][2], lacc[2];
        lacc[0] = zero; lacc[1] = zero;
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) { o[jd][0] = zero; o[jd][1] = zero; }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            frag_ab p0, p1;
            __builtin_memcpy(&p0, pb[0][t], 16);
            __builtin_memcpy(&p1, pb[1][t], 16);
            lacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p0, lacc[0], 0, 0, 0);
            lacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p1, lacc[1], 0, 0, 0);
#pragma unroll
            for (int jd = 0; jd < 4; ++jd) {
                uint4 vv = *(const uint4*)(vbase + jd * 16 * (Q_LDV * 2) + t * 64);
                if (t == 2 && lg >= 2) vv = make_uint4(0u, 0u, 0u, 0u);      // key slots 80 .. 95 lie behind the row: P is 0 there, the bytes must be too
                const frag_ab vf = *(frag_ab*)&vv;
                o[jd][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, p0, o[jd][0], 0, 0, 0);
                o[jd][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, p1, o[jd][1], 0, 0, 0);
            }
        }
        // O^T / l in the accumulator layout -> the wave's patch [32 queries][64 d] fp32 -> row-major, 8 lanes x 8 columns per query, 16-byte stores
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const float inv = 1.0f / lacc[ii][0];
#pragma unroll
            for (int jd = 0; jd < 4; ++jd)
                *(f32x4*)(stg + (ii * 16 + l15) * Q_SR + (jd * 16 + 4 * lg) * 4) = o[jd][ii] * inv;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + (lane >> 3), cg = lane & 7;
            const float4 v0 = *(const float4*)(stg + row * Q_SR + cg * 32), v1 = *(const float4*)(stg + row * Q_SR + cg * 32 + 16);
            uint4 v;
            v.x = pack_bf2(v0.x, v0.y); v.y = pack_bf2(v0.z, v0.w); v.z = pack_bf2(v1.x, v1.y); v.w = pack_bf2(v1.z, v1.w);
            const int m = m0 + ih * 32 + row;
            if (m < p.M) *(uint4*)(Ob + (int64_t)m * qp.ldo + cg * 8) = v;
        }
    }
    if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
}

}  // namespace

int launch_qattn(Params& p, const QAExtra& x, int batch, hipStream_t st) {
    if (p.n_trans_begin >= 0 || p.epilogue != TMIX_EPI_NONE || p.R || p.rgb || p.stats_out || p.f8copy || p.cs_out || p.scaleA)
        TMIX_FAIL(TMIX_EINVAL, "gemm_q_cross_attn: the projection takes a bias and a folded LayerNorm only (no residual / activation / transposed region / statistics / fp8)");
    if ((p.N % Q_BN) || x.rows_per_image <= 0 || (x.rows_per_image % Q_BM) || (p.M % x.rows_per_image))
        TMIX_FAIL(TMIX_ESHAPE, "gemm_q_cross_attn: N=%d must be a multiple of %d (five heads per tile) and M=%d a multiple of rows_per_image=%d, itself a multiple of %d", p.N, Q_BN, p.M, x.rows_per_image, Q_BM);
    if (x.Skv <= 0 || x.Skv > Q_LDV || x.ldvt != Q_LDV) TMIX_FAIL(TMIX_ESHAPE, "gemm_q_cross_attn: Skv=%d must be <= %d and ldvt=%lld == %d", x.Skv, Q_LDV, (long long)x.ldvt, Q_LDV);
    if (!x.K || !x.Vt || !x.O || !aligned16(x.Vt) || (((uintptr_t)x.K) & 7) || (x.ldk % 4) || (x.strideK % 4) || (x.strideVt % 8) || !aligned16(x.O) || (x.ldo % 8))
        TMIX_FAIL(TMIX_EALIGN, "gemm_q_cross_attn: K (8-byte), V^T / O (16-byte) alignment");
    constexpr int SMEM = Q_RING + (Q_BM + Q_BN) * 16 + Q_BM * 4 + Q_BN * 4;
    static_assert(SMEM <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_qattn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    QAParams q;
    p.tiles_m = (p.M + Q_BM - 1) / Q_BM; p.tiles_n = p.N / Q_BN;
    p.group_m = 16;
    // the kernels remap the LINEAR workgroup id over the whole (tiles, slices) grid in 32-bit arithmetic (common.h xcd_remap_grid)
    if ((int64_t)p.tiles_m * p.tiles_n * batch > 0x7fffffffLL) TMIX_FAIL(TMIX_ESHAPE, "gemm: %lld x %d workgroups exceed the 32-bit linear grid id", (long long)p.tiles_m * p.tiles_n, batch);
    dim3 grid(p.tiles_m * p.tiles_n, batch, 1);
    p.prof = tmix_prof_take(&p.prof_detail);
    tmix_prefetch_take(&p.pf, &p.pf_bytes);
    { const long long nthr = (long long)grid.x * grid.y * Q_LW * 64, lines = (p.pf_bytes + 127) >> 7;
      p.pf_per = p.pf ? (int)((lines + nthr - 1) / nthr) : 0; }
    q.g = p;
    q.Kc = (const bf16_t*)x.K; q.ldk = x.ldk; q.strideK = x.strideK;
    q.Vt = (const bf16_t*)x.Vt; q.ldvt = x.ldvt; q.strideVt = x.strideVt;
    q.O = (bf16_t*)x.O; q.ldo = x.ldo;
    q.rows_per_image = x.rows_per_image; q.Skv = x.Skv; q.scale_log2e = x.scale * 1.4426950408889634f;
    gemm_qattn_kernel<<<grid, (Q_NW + Q_LW) * 64, SMEM, st>>>(q);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

}  // namespace tmix_gemm
