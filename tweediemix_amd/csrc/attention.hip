// attention.hip -- flash attention forward for head_dim 64 on gfx950 (bf16 in/out, fp32 softmax).
//
// Replaces the explicit softmax(QK^T)V of utils_custom.py:93-103 / utils_lora.py:101-111 (which
// materialises [B*heads, S, S]) and xformers' attn1 kernel.  One workgroup = 4 waves = 128 query rows
// of one (batch, head); K and V^T tiles of 64 keys are staged HBM->LDS by LDS-DMA (global_load_lds,
// a ring of four tiles, XOR-swizzled through the source address) and shared by the 4 waves.
//
// Register-only softmax: the scores are computed TRANSPOSED, S^T = K Q^T (MFMA A operand = K rows,
// B operand = Q rows), so a lane holds 4 keys x 1 query per 16x16 fragment and the row reductions
// are in-lane plus two cross-lane steps.  The P^T fragments then feed O^T = V^T P^T directly as the
// MFMA B operand: the contraction index (key) may be permuted freely as long as both operands agree,
// so the K rows are staged in the order that makes each lane's own P values the operand it needs
// (key slot (g,j) of k-step p <-> key 32p+8g+j), and V arrives already transposed from the QKV
// projection's epilogue (tmix_gemm_bf16 n_trans_begin) so its fragments are plain 16-byte reads.
//
// MFMA roofline: 4*Sq*Skv*64 flops per (batch, head).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

// TMIX_ATTN_ABL (dev builds under tools/ab/ only): ablations that locate the bound of the tile loop -- bit 0: no exp2 (a multiply
// instead), bit 1: no PV / row-sum MFMAs, bit 2: no QK^T MFMAs, bit 3: no LDS-DMA inside the loop (the ring is re-read), bit 4: no maximum test per tile.
#ifndef TMIX_ATTN_ABL
#define TMIX_ATTN_ABL 0
#endif
constexpr int AABL = TMIX_ATTN_ABL;

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int QB = 128;          // query rows per 4-wave workgroup (32 per wave); the 8-wave form covers 256
constexpr int KB = 64;           // keys per tile
constexpr int TILE = KB * 64 * 2;    // 8 KiB (K tile or V^T tile)
constexpr int STAGE = 2 * TILE;

struct AttnParams {
    const bf16_t* Q; int64_t ldq, strideQ;
    const bf16_t* K; int64_t ldk, strideK;
    const bf16_t* Vt; int64_t ldvt, strideVt;
    bf16_t* O; int64_t ldo, strideO;
    int B, H, Sq, Skv, nq;
    float scale_log2e;
    unsigned long long* prof; int prof_detail;   // in-situ timing slot (common.h) or NULL
    // tmix_attn_fwd_f8: the output leaves as e4m3 bytes [B*Sq][ldo8] with one E8M0 scale per (row, 32 columns) in the k-block-major form
    // [C/32][ldSc] -- the block-scaled A operand of the out-projection (tmix_gemm_fp8, TMIX_F8_A_BLOCK_SCALES); O is not written then
    unsigned char* O8; int64_t ldo8; unsigned char* Sc; int64_t ldSc;
    // key-split tail (attn_fwd_pipe_kernel): the first n_full workgroups take whole (batch, head, query block) items, the items behind them are cut into
    // 1 << lsplit key ranges of tpp tiles each (one workgroup per range); partial results meet in ws, the item's last arriver (ticket) merges and stores
    int n_full, lsplit, tpp; float* ws; unsigned* tickets;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// cross-lane reductions over the 4 lane groups (lane ^ 16, lane ^ 32) with gfx950's VALU row/half swaps
// instead of ds_bpermute: no LDS round trip (8 dependent ~100-cycle LDS latencies per tile otherwise).
__device__ __forceinline__ float xor16_max(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sum(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// one query row of a head (64 columns over the four lanes fg = 0..3 of a 16-lane row block: lane holds columns df * 16 + fg * 4 .. + 3, df = 0..3)
// as e4m3 with one scale per 32 columns: what a torch MX quantiser makes of the bf16-rounded row (the values are rounded to bf16 first, so the
// bytes equal a quantiser pass over the bf16 tensor the plain kernel would have written)
__device__ __forceinline__ void store_row_f8(const AttnParams& p, int64_t row, int h, int fg, const float (&v)[4][4], bool ok) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        float f[8], am = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f[k] = (float)(__bf16)v[2 * blk + (k >> 2)][k & 3];
            am = fmaxf(am, fabsf(f[k]));
        }
        am = xor16_max(am); am = xor32_max(am);
        const int e = e8m0_for_amax(am);
        const float inv = exp2_neg_int(e);
        int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false); q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, q0, true);
        int q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, q1, true);
        if (ok) {
            unsigned char* dst = p.O8 + row * p.ldo8 + h * 64 + blk * 32 + fg * 4;
            *(int*)dst = q0; *(int*)(dst + 16) = q1;
            if (fg == 0) p.Sc[(int64_t)(h * 2 + blk) * p.ldSc + row] = (unsigned char)(e + 127);
        }
    }
}

// 16-byte store / load at the device's coherence point (what an agent-scope atomic store / load does on gfx950, sc1, at four times the width): partial
// results that workgroups on DIFFERENT XCDs (different L2s) exchange inside one launch.  The load is asynchronous: s_waitcnt vmcnt before the value is used.
constexpr int SPLIT_MAX = 4;         // key ranges per item, at most
constexpr int SPLIT_VEC = 9;         // f32x4 per lane of a key range's partial result: o[4][2] + (reference maxima, row sums)
__device__ __forceinline__ void st_coherent(f32x4* dst, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld_coherent(const f32x4* src) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(src) : "memory"); return v; }

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return *(uint32_t*)&v;
}

// ---- self-attention, software-pipelined over KV tiles.
// Round 2's kernel ran a tile as QK^T MFMAs -> softmax VALU -> PV MFMAs, each stage waiting for the one before (matrix pipe ~25 %
// busy; ablations: tools/jobs/r3j_attn_abl.sh, r3m.sh).  Here one loop iteration works on TWO tiles: the QK^T MFMAs of tile t are issued with the exp2 / bf16-pack arithmetic of
// tile t-1 between them (one packed word = two exp2 + one convert behind every MFMA), and the PV MFMAs of tile t-1 with the
// maximum search of tile t between them -- MFMAs execute asynchronously, so the VALU work rides in their shadow inside ONE wave.
// The barrier that hands over tile t+1 sits between the two halves; the K fragments of tile t+1 and the LDS-DMA of tile t+3
// are requested under the PV MFMAs, the V^T fragments of tile t-1 under the QK^T MFMAs.  Ring of 4 tiles (t-1 .. t+2), 64 KiB.
constexpr int NSP = 4;
constexpr int SMEM_P = NSP * STAGE;

__global__ void __launch_bounds__(256, 2) attn_fwd_pipe_kernel(const AttnParams p) {
#ifndef TMIX_NO_KERNARG_TOUCH
    kernarg_touch<(int)sizeof(AttnParams)>();
#endif
    constexpr int LOADS = 4;             // LDS-DMA instructions per wave per tile (16 KiB tile, 1 KiB per instruction)
    constexpr int NR = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, blockIdx.x == 0, p.prof_detail);

    // workgroups are dispatched in id order and dealt to the XCDs round robin: ids < n_full are whole items (a contiguous run per XCD: the query blocks of a
    // head share its K / V in one L2); the ids behind them are the key ranges of the remaining items, all ranges of an item on ONE XCD
    int bid, part = 0;
    if ((int)blockIdx.x < p.n_full) bid = xcd_remap(blockIdx.x, p.n_full);
    else {
        const int j = (int)blockIdx.x - p.n_full, k = j >> 3, n_tail8 = ((int)gridDim.x - p.n_full) >> (3 + p.lsplit);
        bid = p.n_full + (j & 7) * n_tail8 + (k >> p.lsplit);
        part = k & ((1 << p.lsplit) - 1);
    }
    const bool split = (int)blockIdx.x >= p.n_full;
    const int bh = bid / p.nq, qt = bid - bh * p.nq;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * QB + w * 32;
    const int key0 = split ? part * p.tpp * KB : 0;              // this workgroup's key range [key0, key0 + skv)
    const int skv = split ? p.tpp * KB : p.Skv;
    const bf16_t* Qb = p.Q + (int64_t)b * p.strideQ + h * 64;
    const bf16_t* Kb = p.K + (int64_t)b * p.strideK + h * 64 + (int64_t)key0 * p.ldk;
    const bf16_t* Vb = p.Vt + (int64_t)b * p.strideVt + (int64_t)h * 64 * p.ldvt + key0;

    // ---- staging: per wave 2 rounds x (8 rows x 8 chunks) for K and for V^T, XOR-swizzled through the source address
    const int lrow = lane >> 3;
    const int schunk = ((lane & 7) ^ lrow) * 8;
    int krow[NR], voff[NR];
    const int ldk = (int)p.ldk, ldvt = (int)p.ldvt;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int rho = (r * 4 + w) * 8 + lrow;            // LDS row 0..63
        const int f = rho >> 4, i = rho & 15;
        krow[r] = 32 * (f >> 1) + 8 * (i >> 2) + 4 * (f & 1) + (i & 3);
        voff[r] = rho * ldvt;
    }
    const int nt = (skv + KB - 1) / KB;
    const int vlim = ldvt - 8 - key0;
    auto stage = [&](int t) {
        char* sK = smem + (t & (NSP - 1)) * STAGE;
        char* sV = sK + TILE;
        const int kv0 = t * KB;
        int c = kv0 + schunk; if (c > vlim) c = vlim;       // fully masked chunk: any finite data
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int off = (r * 4 + w) * 1024;
            int key = kv0 + krow[r]; if (key > skv - 1) key = skv - 1;
            glds16(Kb + (unsigned)(key * ldk + schunk), sK + off);
            glds16(Vb + (unsigned)(voff[r] + c), sV + off);
        }
    };
#pragma unroll
    for (int t = 0; t < NSP - 1; ++t)
        if (t < nt) stage(t);

    // ---- Q fragments (B operand of S^T = K Q^T), scale * log2(e) folded in; loaded while the first tiles are in flight
    frag_ab qf[2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        int q = q0 + qi * 16 + fr; if (q > p.Sq - 1) q = p.Sq - 1;
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
            const frag_ab raw = *(const frag_ab*)(Qb + (int64_t)q * p.ldq + ds * 32 + fg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qi][ds][j] = (__bf16)((float)raw[j] * p.scale_log2e);
        }
    }

    f32x4 o[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    constexpr float THR = 8.0f;
    f32x4 negm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 lacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    frag_ab ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    bool first = true;

    frag_ab kf[2][4];                  // K fragments of the tile whose QK^T comes next: [d half][key fragment]; half 0 is read
                                       // under the previous PV MFMAs, half 1 under the first QK^T MFMAs (16 registers fewer across the barrier)
    frag_ab vf[2][4];                  // V^T fragments of the tile whose PV comes next: [k-step][d fragment]
    uint32_t pb[2][2][4];              // packed P^T of that tile: [qi][k-step] = B operand of O^T = V^T P^T
    const int swk0 = ((0 * 4 + fg) ^ (fr & 7)) << 4, swk1 = ((1 * 4 + fg) ^ (fr & 7)) << 4;
    auto read_k = [&](int t) {
        const char* sK = smem + (t & (NSP - 1)) * STAGE;
#pragma unroll
        for (int f = 0; f < 4; ++f) kf[0][f] = *(const frag_ab*)(sK + (f * 16 + fr) * 128 + swk0);
    };

    // first half of an iteration: QK^T of tile t into sc (CUR) with exp2 + pack of tile t-1 (sp -> pb) and its V^T fragment reads
    // spread between the MFMAs (PREV); issue order pinned
    auto half1 = [&](int t, f32x4 (&sc)[4][2], f32x4 (&sp)[4][2], auto cur_tag, auto prev_tag) {
        constexpr bool CUR = decltype(cur_tag)::value, PREV = decltype(prev_tag)::value;
        const char* sV = smem + ((t - 1) & (NSP - 1)) * STAGE + TILE;
        const char* sKc = smem + (t & (NSP - 1)) * STAGE;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            // MFMA n: d half n >> 3, key fragment (n >> 1) & 3, query fragment n & 1   (the two d halves of a score are 8 MFMAs apart)
            const int dh = n >> 3, f = (n >> 1) & 3, qi = n & 1;
            if constexpr (CUR) {
                if constexpr (AABL & 4) { if (dh == 0) sc[f][qi] = negm[qi]; asm volatile("" :: "v"(kf[dh][f])); }
                else sc[f][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[dh][f], qf[qi][dh], dh ? sc[f][qi] : negm[qi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (CUR) { if (n < 4) kf[1][n] = *(const frag_ab*)(sKc + (n * 16 + fr) * 128 + swk1); }
            if constexpr (PREV) {
                // packed word n of P^T(t-1): scores (f, qi, 2 hh) and (f, qi, 2 hh + 1) with f = n >> 2, qi = (n >> 1) & 1, hh = n & 1
                const int pf = n >> 2, pq = (n >> 1) & 1, hh = n & 1;
                const float e0 = (AABL & 1) ? sp[pf][pq][2 * hh] * 0.001f : __builtin_amdgcn_exp2f(sp[pf][pq][2 * hh]);
                const float e1 = (AABL & 1) ? sp[pf][pq][2 * hh + 1] * 0.001f : __builtin_amdgcn_exp2f(sp[pf][pq][2 * hh + 1]);
                pb[pq][pf >> 1][(pf & 1) * 2 + hh] = pk_bf16(e0, e1);
                asm volatile("" : "+v"(pb[pq][pf >> 1][(pf & 1) * 2 + hh]));      // computed HERE (LLVM otherwise sinks it to its first use, behind the barrier)
                if (n >= 4 && n < 12) { const int m = n - 4; vf[m >> 2][m & 3] = *(const frag_ab*)(sV + ((m & 3) * 16 + fr) * 128 + ((((m >> 2) * 4 + fg) ^ (fr & 7)) << 4)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // second half: PV (and row sums) of tile t-1 (PREV) with the maximum search of tile t (CUR) between the MFMAs; the K fragments
    // of tile t+1 are requested here too.  Then, rarely, the running maximum moves (deferred rescale: P <= 2^THR stays well inside bf16 / fp32 range).
    auto half2 = [&](int t, f32x4 (&sc)[4][2], auto cur_tag, auto prev_tag, auto mask_tag) {
        constexpr bool CUR = decltype(cur_tag)::value, PREV = decltype(prev_tag)::value, MASK = decltype(mask_tag)::value;
        if constexpr (CUR && MASK) {
            const int kv0 = t * KB;
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + 32 * (f >> 1) + 8 * fg + 4 * (f & 1) + r >= skv) sc[f][qi][r] = -INFINITY;
        }
        int im = 0x80000000;
        const char* sKn = smem + ((t + 1) & (NSP - 1)) * STAGE;
#pragma unroll
        for (int n = 0; n < 20; ++n) {
            if constexpr (PREV) {
                const int ps = n / 10, m = n % 10;       // per k-step: two row-sum MFMAs, then 4 d fragments x 2 query fragments
                frag_ab p0, p1;
                __builtin_memcpy(&p0, pb[0][ps], 16);
                __builtin_memcpy(&p1, pb[1][ps], 16);
                if constexpr (AABL & 2) asm volatile("" :: "v"(p0), "v"(p1), "v"(vf[ps][(m >> 1) & 3]));
                else if (m == 0) lacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p0, lacc[0], 0, 0, 0);
                else if (m == 1) lacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p1, lacc[1], 0, 0, 0);
                else if (m & 1) o[(m - 2) >> 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ps][(m - 2) >> 1], p1, o[(m - 2) >> 1][1], 0, 0, 0);
                else o[(m - 2) >> 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ps][(m - 2) >> 1], p0, o[(m - 2) >> 1][0], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (CUR) {
                if (n < 8) {           // integer order = float order for "is any score > THR (> 0)"
                    const int f = n >> 1, qi = n & 1;
                    im = max(max(im, max(__float_as_int(sc[f][qi][0]), __float_as_int(sc[f][qi][1]))),
                             max(__float_as_int(sc[f][qi][2]), __float_as_int(sc[f][qi][3])));
                } else if (n < 12) {   // K fragments (first d half) of the next tile (a stale slot behind the last tile: never used)
                    kf[0][n - 8] = *(const frag_ab*)(sKn + ((n - 8) * 16 + fr) * 128 + swk0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (CUR) {
            if ((AABL & 16) ? first : (first || __any(im > __float_as_int(THR)))) {
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    float m0 = fmaxf(fmaxf(sc[0][qi][0], sc[0][qi][1]), fmaxf(sc[0][qi][2], sc[0][qi][3]));
#pragma unroll
                    for (int f = 1; f < 4; ++f) m0 = fmaxf(fmaxf(m0, fmaxf(sc[f][qi][0], sc[f][qi][1])), fmaxf(sc[f][qi][2], sc[f][qi][3]));
                    float delta = xor32_max(xor16_max(m0));
                    if (!first) delta = fmaxf(delta, 0.f);               // never lower an established maximum
                    const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int f = 0; f < 4; ++f) sc[f][qi] = sc[f][qi] - delta;
#pragma unroll
                    for (int df = 0; df < 4; ++df) o[df][qi] *= alpha;
                    lacc[qi] *= alpha;
                    negm[qi] = negm[qi] - delta;
                }
                first = false;
            }
        }
    };
    // hand-over between the halves of iteration t: this wave's LDS reads are in registers, tile t+1 has landed for everybody,
    // tile t-1's ring slot is free -> tile t+3 is requested into it
    auto sync = [&](int t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + 2 < nt) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if (!(AABL & 8) && t + 3 < nt) stage(t + 3);
    };

    f32x4 sA[4][2], sB[4][2];
    const bool tail_masked = (skv % KB) != 0;
    using T_ = std::true_type; using F_ = std::false_type;
    if (nt >= NSP - 1) wait_vmcnt<(NSP - 2) * LOADS>(); else if (nt == 2) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (prof_on) pt1 = prof_now();
    read_k(0);
    // iteration 0 has no previous tile
    half1(0, sA, sB, T_{}, F_{});
    sync(0);
    if (nt == 1 && tail_masked) half2(0, sA, T_{}, F_{}, T_{}); else half2(0, sA, T_{}, F_{}, F_{});
    int t = 1;
    for (; t + 1 < nt; t += 2) {       // two iterations per trip: the score registers swap roles (sA <-> sB) without copies
        half1(t, sB, sA, T_{}, T_{});
        sync(t);
        half2(t, sB, T_{}, T_{}, F_{});
        half1(t + 1, sA, sB, T_{}, T_{});
        sync(t + 1);
        if (t + 2 == nt && tail_masked) half2(t + 1, sA, T_{}, T_{}, T_{}); else half2(t + 1, sA, T_{}, T_{}, F_{});
    }
    if (t < nt) {                      // odd tile left: it is the last one
        half1(t, sB, sA, T_{}, T_{});
        sync(t);
        if (tail_masked) half2(t, sB, T_{}, T_{}, T_{}); else half2(t, sB, T_{}, T_{}, F_{});
        // drain: exp2 / pack and PV of the last tile (scores in sB)
        half1(t + 1, sA, sB, F_{}, T_{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        half2(t + 1, sA, F_{}, T_{}, F_{});
    } else {                           // the last tile's scores are in sA
        half1(nt, sB, sA, F_{}, T_{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        half2(nt, sB, F_{}, T_{}, F_{});
    }

    if (prof_on) pt2 = prof_now();
    if (split) {
        // publish this key range's (o, reference maximum, row sum) in the lane layout it is held in (nine 16-byte vectors per lane, written through to the
        // coherence point), take a ticket; the item's last arriver re-reads ALL ranges in range order (so the sum does not depend on who arrives last)
        const int nparts = 1 << p.lsplit, itail = bid - p.n_full;
        f32x4* wsb = (f32x4*)p.ws + ((int64_t)(itail * nparts) * 4 + w) * (SPLIT_VEC * 64) + lane;      // range r: + r * 4 * SPLIT_VEC * 64
        {
            f32x4* dst = wsb + (int64_t)part * 4 * (SPLIT_VEC * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) st_coherent(dst + i * 64, o[i >> 1][i & 1]);
            st_coherent(dst + 8 * 64, (f32x4){negm[0][0], negm[1][0], lacc[0][0], lacc[1][0]});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = (unsigned*)smem;
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.tickets + itail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old == (unsigned)(nparts - 1);
            if (last) __hip_atomic_store(p.tickets + itail, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
            *flag = last;
        }
        __syncthreads();
        if (!*flag) { if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2); return; }
        f32x4 st[SPLIT_MAX];
#pragma unroll
        for (int r = 0; r < SPLIT_MAX; ++r) if (r < nparts) st[r] = ld_coherent(wsb + (int64_t)r * 4 * (SPLIT_VEC * 64) + 8 * 64);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
        for (int r = 0; r < SPLIT_MAX; ++r) if (r < nparts) { asm volatile("" : "+v"(st[r])); m0 = fmaxf(m0, -st[r][0]); m1 = fmaxf(m1, -st[r][1]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int r = 0; r < SPLIT_MAX; ++r) {
            if (r >= nparts) break;
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ld_coherent(wsb + (int64_t)r * 4 * (SPLIT_VEC * 64) + i * 64);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float w0 = __builtin_amdgcn_exp2f(-st[r][0] - m0), w1 = __builtin_amdgcn_exp2f(-st[r][1] - m1);
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(v[i])); o[i >> 1][i & 1] += v[i] * ((i & 1) ? w1 : w0); }
            l0 += st[r][2] * w0; l1 += st[r][3] * w1;
        }
        lacc[0][0] = l0; lacc[1][0] = l1;
    }
    if (p.O8) {
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int q = q0 + qi * 16 + fr;
            const float inv = 1.0f / lacc[qi][0];
            float v[4][4];
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[df][r] = o[df][qi][r] * inv;
            store_row_f8(p, (int64_t)b * p.Sq + min(q, p.Sq - 1), h, fg, v, q < p.Sq);
        }
        if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
        return;
    }
    bf16_t* Ob = p.O + (int64_t)b * p.strideO + h * 64;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int q = q0 + qi * 16 + fr;
        if (q >= p.Sq) continue;
        const float inv = 1.0f / lacc[qi][0];
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            uint2 v;
            v.x = pk_bf16(o[df][qi][0] * inv, o[df][qi][1] * inv);
            v.y = pk_bf16(o[df][qi][2] * inv, o[df][qi][3] * inv);
            *(uint2*)(Ob + (int64_t)q * p.ldo + df * 16 + fg * 4) = v;
        }
    }
    if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
}

// ---- cross-attention against a SHORT key set (the 77 prompt tokens: diffusers attn2, utils_custom.py:93-103): K and V^T of one
// (batch, head) are 2 x 10 KB, so every wave keeps ALL of them in registers as MFMA operands (no LDS, no barrier, no online
// softmax: one exact maximum per query) and walks 64 queries in two blocks of 32.  The general kernel above spent 27 us per
// launch on this shape (a 3-stage LDS ring and a barrier per 64-key tile for 1.2 tiles of work); the floor here is streaming
// Q in and O out once.  Key slots follow the same permutation as above, so P^T feeds the PV MFMA from registers.
constexpr int SK_MAX = 96;           // key slots (3 MFMA k-steps of 32)
constexpr int SQW = 64;              // queries per wave

// The waves of a workgroup share nothing (no LDS, no barrier), so the work is dealt out per WAVE: wave gw of the launch takes queries
// [64 (gw % nq), +64) of (batch, head) gw / nq, and the host picks the workgroup size -- 4 or 5 waves -- that spreads the waves evenly
// over the 256 CUs (B = 4, 20 heads, 1024 queries: 1280 waves = 256 workgroups of five, one per CU, where 320 workgroups of four ran
// 1.25 rounds).
__global__ void __launch_bounds__(320, 2) attn_small_kernel(const AttnParams p) {
#ifndef TMIX_NO_KERNARG_TOUCH
    kernarg_touch<(int)sizeof(AttnParams)>();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, blockIdx.x == 0, p.prof_detail);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int gw = bid * (int)(blockDim.x >> 6) + w;           // p.nq = 64-query blocks per (batch, head)
    const int bh = gw / p.nq, qt = gw - bh * p.nq;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * SQW;
    if (b >= p.B) {                                            // surplus waves of the last workgroup
        if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt0, pt0);
        return;
    }
    const bf16_t* Qb = p.Q + (int64_t)b * p.strideQ + h * 64;
    const bf16_t* Kb = p.K + (int64_t)b * p.strideK + h * 64;
    const bf16_t* Vb = p.Vt + (int64_t)b * p.strideVt + (int64_t)h * 64 * p.ldvt;
    bf16_t* Ob = p.O + (int64_t)b * p.strideO + h * 64;
    const int ldvt = (int)p.ldvt;

    frag_ab kf[6][2], vf[3][4];
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        int key = 32 * (f >> 1) + 8 * (fr >> 2) + 4 * (f & 1) + (fr & 3); if (key > p.Skv - 1) key = p.Skv - 1;
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) kf[f][ds] = *(const frag_ab*)(Kb + (int64_t)key * p.ldk + ds * 32 + fg * 8);
    }
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        int c = 32 * ps + fg * 8; if (c > ldvt - 8) c = ldvt - 8;          // fully masked chunk: any finite data
#pragma unroll
        for (int df = 0; df < 4; ++df) vf[ps][df] = *(const frag_ab*)(Vb + (df * 16 + fr) * ldvt + c);
    }
    frag_ab ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    // the queries of BOTH 32-query blocks are requested here, in the same memory round trip as K / V^T: loaded at the top of each block they cost
    // the second block a round trip of its own, in a kernel whose whole run is a few of them
    frag_ab qraw[SQW / 32][2][2];
#pragma unroll
    for (int blk = 0; blk < SQW / 32; ++blk)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            int q = q0 + blk * 32 + qi * 16 + fr; if (q > p.Sq - 1) q = p.Sq - 1;
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) qraw[blk][qi][ds] = *(const frag_ab*)(Qb + (int64_t)q * p.ldq + ds * 32 + fg * 8);
        }
    if (prof_on) pt1 = prof_now();

#pragma unroll
    for (int blk = 0; blk < SQW / 32; ++blk) {
        const int qb = q0 + blk * 32;
        if (qb >= p.Sq) break;
        frag_ab qf[2][2];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                const frag_ab raw = qraw[blk][qi][ds];
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qi][ds][j] = (__bf16)((float)raw[j] * p.scale_log2e);
            }
        }
        f32x4 s[6][2];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 6; ++f)
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                s[f][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][0], qf[qi][0], zero, 0, 0, 0);
                s[f][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[f][1], qf[qi][1], s[f][qi], 0, 0, 0);
            }
        uint32_t pb[2][3][4];
        f32x4 o[4][2], lacc[2];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            float m = -INFINITY;
#pragma unroll
            for (int f = 0; f < 6; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (32 * (f >> 1) + 8 * fg + 4 * (f & 1) + r >= p.Skv) s[f][qi][r] = -INFINITY;
                    m = fmaxf(m, s[f][qi][r]);
                }
            m = xor32_max(xor16_max(m));                               // over the 4 lane groups of this query column
#pragma unroll
            for (int f = 0; f < 6; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[f][qi][r] = __builtin_amdgcn_exp2f(s[f][qi][r] - m);
#pragma unroll
            for (int ps = 0; ps < 3; ++ps) {
                pb[qi][ps][0] = pk_bf16(s[2 * ps][qi][0], s[2 * ps][qi][1]);
                pb[qi][ps][1] = pk_bf16(s[2 * ps][qi][2], s[2 * ps][qi][3]);
                pb[qi][ps][2] = pk_bf16(s[2 * ps + 1][qi][0], s[2 * ps + 1][qi][1]);
                pb[qi][ps][3] = pk_bf16(s[2 * ps + 1][qi][2], s[2 * ps + 1][qi][3]);
            }
            lacc[qi] = zero;
#pragma unroll
            for (int df = 0; df < 4; ++df) o[df][qi] = zero;
        }
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
            frag_ab p0, p1;
            __builtin_memcpy(&p0, pb[0][ps], 16);
            __builtin_memcpy(&p1, pb[1][ps], 16);
            lacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p0, lacc[0], 0, 0, 0);
            lacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p1, lacc[1], 0, 0, 0);
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                o[df][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ps][df], p0, o[df][0], 0, 0, 0);
                o[df][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ps][df], p1, o[df][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int q = qb + qi * 16 + fr;
            const float inv = 1.0f / lacc[qi][0];
            if (p.O8) {                                    // wave-uniform
                float v[4][4];
#pragma unroll
                for (int df = 0; df < 4; ++df)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[df][r] = o[df][qi][r] * inv;
                store_row_f8(p, (int64_t)b * p.Sq + min(q, p.Sq - 1), h, fg, v, q < p.Sq);
                continue;
            }
            if (q >= p.Sq) continue;
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                uint2 v;
                v.x = pk_bf16(o[df][qi][0] * inv, o[df][qi][1] * inv);
                v.y = pk_bf16(o[df][qi][2] * inv, o[df][qi][3] * inv);
                *(uint2*)(Ob + (int64_t)q * p.ldo + df * 16 + fg * 4) = v;
            }
        }
    }
    if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt1);
}

}  // namespace

// Key-split tail: B * H * ceil(Sq / 128) items on 512 workgroup slots (two per CU).  When a last, partly filled round remains (SDXL: 640 items at S = 1024,
// 1280 at S = 4096) its r items are cut into 512 / r key ranges, one workgroup each, so that the last round is 1 / split as long and fills every slot.
struct AttnSplit { int n_full, lsplit, tpp, n_tail; int64_t ws_bytes; };
// (SLOTS and the tail's tile -> XCD mapping -- j >> 3, j & 7 in the kernel -- are the MI355X's: 256 CUs x 2 resident workgroups, 8 XCDs.  On a part with another CU
// count "all ranges of an item on one XCD" and "the split fills the last round" no longer hold, so the split switches itself off there; the library refuses non-gfx950
// devices anyway, tmix_check_device.  Asked at launch time: tmix_attn_split_ws_bytes is a pure function of the shape.)
static bool attn_split_device_ok() {
    static int ok = -1;
    if (ok < 0) {
        int dev = 0; hipDeviceProp_t prop;
        ok = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount * 2 == 512) ? 1 : 0;
    }
    return ok == 1;
}
static bool attn_split_plan(int B, int H, int Sq, int Skv, AttnSplit& sp) {
    constexpr int SLOTS = 512, TICKET_BYTES = 4096;
    if (Skv <= SK_MAX || (Skv % KB) || tmix_env(TMIX_ENV_ATTN_NO_SPLIT)) return false;
    const int64_t items = (int64_t)((Sq + QB - 1) / QB) * B * H;
    const int nt = Skv / KB;
    const int r = (int)(items % SLOTS);
    if (items <= SLOTS || r == 0 || (r & 7) || (SLOTS % r)) return false;
    int split = SLOTS / r, lsplit = 0;
    if (split > SPLIT_MAX) split = SPLIT_MAX;
    while ((1 << lsplit) < split) ++lsplit;
    if ((1 << lsplit) != split || (nt % split) || nt / split < 2 || r > TICKET_BYTES / 4) return false;
    sp.n_full = (int)(items - r); sp.lsplit = lsplit; sp.tpp = nt / split; sp.n_tail = r;
    sp.ws_bytes = TICKET_BYTES + (int64_t)r * split * 4 * SPLIT_VEC * 64 * 16;
    return true;
}

extern "C" int64_t tmix_attn_split_ws_bytes(int B, int H, int Sq, int Skv) {
    AttnSplit sp;
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) return 0;
    return attn_split_plan(B, H, Sq, Skv, sp) ? sp.ws_bytes : 0;
}

static int attn_entry(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                      const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                      void* O8, int64_t ldo8, void* Sc, int64_t ldSc,
                      int B, int H, int Sq, int Skv, float scale, void* stream, void* ws = nullptr, int64_t ws_bytes = 0) {
    if (!Q || !K || !Vt || (!O && !O8)) TMIX_FAIL(TMIX_EINVAL, "attn: null pointer");
    if (O8) {
        if (!Sc || (ldo8 % 4) || ldo8 < (int64_t)H * 64 || ldSc < (int64_t)B * Sq || (((uintptr_t)O8) & 3))
            TMIX_FAIL(TMIX_EINVAL, "attn_f8: needs the scale array [H*2][ldSc >= B*Sq] and 4-byte aligned e4m3 rows of ldo8 >= H*64 bytes");
        O = (void*)Q; ldo = 4; strideO = 4;        // (the bf16-output checks below do not apply; p.O is never written)
    }
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) TMIX_FAIL(TMIX_ESHAPE, "attn: empty problem B=%d H=%d Sq=%d Skv=%d", B, H, Sq, Skv);
    if ((int64_t)Skv * ldk >= (1ll << 31) || 64 * ldvt >= (1ll << 31)) TMIX_FAIL(TMIX_ESHAPE, "attn: per-head K/V extent exceeds 32-bit offsets");
    if ((ldq % 8) || (ldk % 8) || (ldvt % 8) || (ldo % 4) || (strideQ % 8) || (strideK % 8) || (strideVt % 8) || (strideO % 4))
        TMIX_FAIL(TMIX_EALIGN, "attn: leading dims / strides must keep 16-byte (Q,K,Vt) / 8-byte (O) alignment");
    if (ldvt < ((Skv + 7) / 8) * 8) TMIX_FAIL(TMIX_ESHAPE, "attn: ldvt=%lld < Skv rounded up to 8", (long long)ldvt);
    if (!aligned16(Q) || !aligned16(K) || !aligned16(Vt) || (((uintptr_t)O) & 7)) TMIX_FAIL(TMIX_EALIGN, "attn: pointer alignment");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_P);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    AttnParams p;
    p.Q = (const bf16_t*)Q; p.ldq = ldq; p.strideQ = strideQ;
    p.K = (const bf16_t*)K; p.ldk = ldk; p.strideK = strideK;
    p.Vt = (const bf16_t*)Vt; p.ldvt = ldvt; p.strideVt = strideVt;
    p.O = (bf16_t*)O; p.ldo = ldo; p.strideO = strideO;
    p.O8 = (unsigned char*)O8; p.ldo8 = ldo8; p.Sc = (unsigned char*)Sc; p.ldSc = ldSc;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.nq = (Sq + QB - 1) / QB;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.prof = tmix_prof_take(&p.prof_detail);
    if (Skv <= SK_MAX && !tmix_env(TMIX_ENV_ATTN_GENERAL)) {          // short key set: K / V^T resident in registers, no LDS
        p.nq = (Sq + SQW - 1) / SQW;
        const int64_t waves = (int64_t)p.nq * B * H;
        // workgroups of five waves when that fills whole rounds of 256 CUs better than four (waves / CU rounded up, then fewer workgroups)
        const int64_t w4 = (waves + 3) / 4, w5 = (waves + 4) / 5;
        const int64_t c4 = ((w4 + 255) / 256) * 4, c5 = ((w5 + 255) / 256) * 5;
        const int nwv = c5 < c4 ? 5 : 4;
        const int64_t nws = nwv == 5 ? w5 : w4;
        if (nws > 0x7fffffff) TMIX_FAIL(TMIX_ESHAPE, "attn: grid too large");
        attn_small_kernel<<<dim3((unsigned)nws), nwv * 64, 0, (hipStream_t)stream>>>(p);
        TMIX_LAUNCH_CHECK();
        return TMIX_OK;
    }
    int64_t nwg = (int64_t)p.nq * B * H;
    if (nwg > 0x7fffffff) TMIX_FAIL(TMIX_ESHAPE, "attn: grid too large");
    p.n_full = (int)nwg; p.lsplit = 0; p.tpp = 0; p.ws = nullptr; p.tickets = nullptr;
    AttnSplit sp;
    if (ws && attn_split_device_ok() && attn_split_plan(B, H, Sq, Skv, sp)) {     // (the size query stays a pure function of the shape; the device is asked here)
        if ((((uintptr_t)ws) & 15) || ws_bytes < sp.ws_bytes)
            TMIX_FAIL(TMIX_EINVAL, "attn: the key-split workspace needs %lld bytes (tmix_attn_split_ws_bytes), 16-byte aligned; got %lld", (long long)sp.ws_bytes, (long long)ws_bytes);
        p.n_full = sp.n_full; p.lsplit = sp.lsplit; p.tpp = sp.tpp;
        p.tickets = (unsigned*)ws; p.ws = (float*)((char*)ws + 4096);
        nwg = (int64_t)sp.n_full + ((int64_t)sp.n_tail << sp.lsplit);
    }
    attn_fwd_pipe_kernel<<<dim3((unsigned)nwg), 256, SMEM_P, (hipStream_t)stream>>>(p);
    if (p.tickets) {      // a launch that did not start leaves no one to reset the tickets: re-zero them (best effort), or every later launch on this workspace would never merge
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) {
            (void)hipMemsetAsync(p.tickets, 0, 4096, (hipStream_t)stream);
            TMIX_FAIL((int)e_, "%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_));
        }
        return TMIX_OK;
    }
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_attn_fwd(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                             const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                             int B, int H, int Sq, int Skv, float scale, void* stream) {
    return attn_entry(Q, ldq, strideQ, K, ldk, strideK, Vt, ldvt, strideVt, O, ldo, strideO, nullptr, 0, nullptr, 0, B, H, Sq, Skv, scale, stream);
}

extern "C" int tmix_attn_fwd_f8(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                                const void* Vt, int64_t ldvt, int64_t strideVt, void* O8, int64_t ldo8, void* scales, int64_t ldScale,
                                int B, int H, int Sq, int Skv, float scale, void* stream) {
    if (!O8) TMIX_FAIL(TMIX_EINVAL, "attn_f8: null output");
    return attn_entry(Q, ldq, strideQ, K, ldk, strideK, Vt, ldvt, strideVt, nullptr, 0, 0, O8, ldo8, scales, ldScale, B, H, Sq, Skv, scale, stream);
}

// The same two with a caller-owned workspace for the key-split tail (tmix_attn_split_ws_bytes; zero-filled once by the caller, left zeroed by every launch;
// one workspace per stream that runs attention).  ws = NULL or a shape that does not split: exactly tmix_attn_fwd / tmix_attn_fwd_f8.
extern "C" int tmix_attn_fwd_ws(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                                const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                                int B, int H, int Sq, int Skv, float scale, void* ws, int64_t ws_bytes, void* stream) {
    return attn_entry(Q, ldq, strideQ, K, ldk, strideK, Vt, ldvt, strideVt, O, ldo, strideO, nullptr, 0, nullptr, 0, B, H, Sq, Skv, scale, stream, ws, ws_bytes);
}

extern "C" int tmix_attn_fwd_f8_ws(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                                   const void* Vt, int64_t ldvt, int64_t strideVt, void* O8, int64_t ldo8, void* scales, int64_t ldScale,
                                   int B, int H, int Sq, int Skv, float scale, void* ws, int64_t ws_bytes, void* stream) {
    if (!O8) TMIX_FAIL(TMIX_EINVAL, "attn_f8: null output");
    return attn_entry(Q, ldq, strideQ, K, ldk, strideK, Vt, ldvt, strideVt, nullptr, 0, 0, O8, ldo8, scales, ldScale, B, H, Sq, Skv, scale, stream, ws, ws_bytes);
}
