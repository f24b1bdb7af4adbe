// attention.hip -- flash attention forward for head_dim 64 on gfx950 (bf16 in/out, fp32 softmax).
//
// Replaces the explicit softmax(QK^T)V of utils_custom.py:93-103 / utils_lora.py:101-111 (which
// materialises [B*heads, S, S]) and xformers' attn1 kernel.  One workgroup = 4 waves = 128 query rows
// of one (batch, head); K and V^T tiles of 64 keys are staged HBM->LDS by LDS-DMA (global_load_lds,
// double buffered, XOR-swizzled through the source address) and shared by the 4 waves.
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

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int QB = 128;          // query rows per 4-wave workgroup (32 per wave); the 8-wave form covers 256
constexpr int KB = 64;           // keys per tile
constexpr int TILE = KB * 64 * 2;    // 8 KiB (K tile or V^T tile)
constexpr int STAGE = 2 * TILE;
constexpr int NS = 3;                // LDS ring depth: NS-1 tiles requested ahead, NS-2 in flight across a barrier
constexpr int SMEM = NS * STAGE;     // 48 KiB -> 3 workgroups per CU

struct AttnParams {
    const bf16_t* Q; int64_t ldq, strideQ;
    const bf16_t* K; int64_t ldk, strideK;
    const bf16_t* Vt; int64_t ldvt, strideVt;
    bf16_t* O; int64_t ldo, strideO;
    int H, Sq, Skv, nq;
    float scale_log2e;
    unsigned long long* prof; int prof_detail;   // in-situ timing slot (common.h) or NULL
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

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return *(uint32_t*)&v;
}

// NWV = 4 (default): 128 query rows per workgroup, two workgroups per CU.  NWV = 8: 256 rows, ONE workgroup per CU -- the same
// eight waves per CU, a K / V^T tile staged once for 256 queries instead of twice for 2 x 128 (half the LDS-DMA instructions
// per wave, half the L2 -> LDS traffic) -- but slower in situ, see tmix_attn_fwd.
template <int NWV>
__global__ void __launch_bounds__(NWV * 64, 2) attn_fwd_kernel(const AttnParams p) {
    constexpr int LOADS = 16 / NWV;      // LDS-DMA instructions per wave per tile (16 KiB tile, 1 KiB per instruction)
    constexpr int NR = 8 / NWV;          // staging rounds per wave and operand
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, blockIdx.x == 0, p.prof_detail);

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = bid / p.nq, qt = bid - bh * p.nq;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * (NWV * 32) + w * 32;

    const bf16_t* Qb = p.Q + (int64_t)b * p.strideQ + h * 64;
    const bf16_t* Kb = p.K + (int64_t)b * p.strideK + h * 64;
    const bf16_t* Vb = p.Vt + (int64_t)b * p.strideVt + (int64_t)h * 64 * p.ldvt;

    // ---- Q fragments (B operand of S^T = K Q^T), kept in registers for the whole kernel
    frag_ab qf[2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        int q = q0 + qi * 16 + fr; if (q > p.Sq - 1) q = p.Sq - 1;
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
            // fold softmax scale * log2(e) into Q once (one extra bf16 rounding of q, none per score)
            const frag_ab raw = *(const frag_ab*)(Qb + (int64_t)q * p.ldq + ds * 32 + fg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qi][ds][j] = (__bf16)((float)raw[j] * p.scale_log2e);
        }
    }

    // ---- staging geometry: per wave 2 rounds x (8 rows x 8 chunks) for K and for V^T
    const int lrow = lane >> 3;
    const int schunk = ((lane & 7) ^ lrow) * 8;
    int krow[NR];                      // key (within tile) whose row lands in this lane's LDS row
    int vrow[NR];                      // d row of V^T
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int rho = (r * NWV + w) * 8 + lrow;          // LDS row 0..63
        const int f = rho >> 4, i = rho & 15;
        krow[r] = 32 * (f >> 1) + 8 * (i >> 2) + 4 * (f & 1) + (i & 3);
        vrow[r] = rho;
    }
    const int nt = (p.Skv + KB - 1) / KB;
    // per-lane source offsets stay 32-bit (elements); the 64-bit bases are wave-uniform
    const int ldk = (int)p.ldk, ldvt = (int)p.ldvt;
    int voff[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) voff[r] = vrow[r] * ldvt;
    auto stage = [&](int buf, int t) {
        char* sK = smem + buf * STAGE;
        char* sV = sK + TILE;
        const int kv0 = t * KB;
        int c = kv0 + schunk; if (c > ldvt - 8) c = ldvt - 8;       // fully masked chunk: any finite data
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int off = (r * NWV + w) * 1024;
            int key = kv0 + krow[r]; if (key > p.Skv - 1) key = p.Skv - 1;
            glds16(Kb + (unsigned)(key * ldk + schunk), sK + off);
            glds16(Vb + (unsigned)(voff[r] + c), sV + off);
        }
    };

    f32x4 o[4][2];                     // O^T accumulators: [d fragment][q fragment]
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // one KV tile: S^T = K Q^T (already in log2 units, accumulated on top of -m so the MFMA does the subtraction),
    // P = exp2(S^T), O^T += V^T P^T.  The running maximum m is only moved when some score exceeds it by more than
    // THR (deferred rescale, P <= 2^THR stays well inside bf16/fp32 range), so the common path has no cross-lane
    // traffic, no per-score subtract and no accumulator rescale.  MASK = tile holds keys >= Skv.
    constexpr float THR = 8.0f;
    f32x4 negm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // -m per query column, replicated for the MFMA C operand
    f32x4 lacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // softmax denominators, accumulated BY THE MATRIX CORE
    frag_ab ones;                                                     // A operand of all 1.0: D[i][q] = sum_k P^T[k][q]
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    bool first = true;
    auto tile = [&](int cur, int kv0, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        const char* sK = smem + cur * STAGE;
        const char* sV = sK + TILE;
        f32x4 s[4][2];
        {
            const int sw0 = ((0 * 4 + fg) ^ (fr & 7)) << 4, sw1 = ((1 * 4 + fg) ^ (fr & 7)) << 4;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const frag_ab k0 = *(const frag_ab*)(sK + (f * 16 + fr) * 128 + sw0);
                s[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[0][0], negm[0], 0, 0, 0);
                s[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[1][0], negm[1], 0, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const frag_ab k1 = *(const frag_ab*)(sK + (f * 16 + fr) * 128 + sw1);
                s[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[0][1], s[f][0], 0, 0, 0);
                s[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[1][1], s[f][1], 0, 0, 0);
            }
        }
        if constexpr (MASK) {
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + 32 * (f >> 1) + 8 * fg + 4 * (f & 1) + r >= p.Skv) s[f][qi][r] = -INFINITY;
        }
        // threshold test on the raw bit patterns: for "is any score > THR (> 0)" signed-integer order equals float
        // order (negative floats are negative ints), and integer max needs no IEEE canonicalisation of MFMA outputs.
        int im = 0x80000000;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                im = max(max(im, max(__float_as_int(s[f][qi][0]), __float_as_int(s[f][qi][1]))),
                         max(__float_as_int(s[f][qi][2]), __float_as_int(s[f][qi][3])));
        if (first || __any(im > __float_as_int(THR))) {
            // move the maximum: delta = row max relative to the old m (over all 4 lane groups of the column)
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                float m0 = fmaxf(fmaxf(s[0][qi][0], s[0][qi][1]), fmaxf(s[0][qi][2], s[0][qi][3]));
#pragma unroll
                for (int f = 1; f < 4; ++f) m0 = fmaxf(fmaxf(m0, fmaxf(s[f][qi][0], s[f][qi][1])), fmaxf(s[f][qi][2], s[f][qi][3]));
                float delta = xor32_max(xor16_max(m0));
                if (!first) delta = fmaxf(delta, 0.f);               // never lower an established maximum
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                for (int f = 0; f < 4; ++f) s[f][qi] = s[f][qi] - delta;
#pragma unroll
                for (int df = 0; df < 4; ++df) o[df][qi] *= alpha;
                lacc[qi] *= alpha;
                negm[qi] = negm[qi] - delta;
            }
            first = false;
        }
        uint32_t pb[2][2][4];          // [qi][k-step] packed bf16x8 = B operand of O^T = V^T P^T
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[f][qi][r] = __builtin_amdgcn_exp2f(s[f][qi][r]);
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                pb[qi][ps][0] = pk_bf16(s[2 * ps][qi][0], s[2 * ps][qi][1]);
                pb[qi][ps][1] = pk_bf16(s[2 * ps][qi][2], s[2 * ps][qi][3]);
                pb[qi][ps][2] = pk_bf16(s[2 * ps + 1][qi][0], s[2 * ps + 1][qi][1]);
                pb[qi][ps][3] = pk_bf16(s[2 * ps + 1][qi][2], s[2 * ps + 1][qi][3]);
            }
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int sw = ((ps * 4 + fg) ^ (fr & 7)) << 4;
            frag_ab p0, p1;
            __builtin_memcpy(&p0, pb[0][ps], 16);
            __builtin_memcpy(&p1, pb[1][ps], 16);
            // row sums of the bf16-rounded P (exactly what multiplies V): one MFMA per (k-step, query fragment)
            lacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p0, lacc[0], 0, 0, 0);
            lacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p1, lacc[1], 0, 0, 0);
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                const frag_ab vf = *(const frag_ab*)(sV + (df * 16 + fr) * 128 + sw);
                o[df][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, p0, o[df][0], 0, 0, 0);
                o[df][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, p1, o[df][1], 0, 0, 0);
            }
        }
    };

    // K/V tiles are small (16 KiB) and a tile's compute is shorter than the L2->LDS latency, so the loop keeps
    // NS-2 tiles in flight across each barrier (counted vmcnt + raw s_barrier, as in the GEMM mainloop).
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) stage(s, s);
    if (nt >= NS - 1) wait_vmcnt<(NS - 2) * LOADS>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (prof_on) pt1 = prof_now();
    int cur = 0, nxt = NS - 1;
    for (int t = 0; t < nt; ++t) {
        const bool more = t + NS - 1 < nt;
        if (more) stage(nxt, t + NS - 1);
        const int kv0 = t * KB;
        if (kv0 + KB > p.Skv) tile(cur, kv0, std::true_type{});
        else                  tile(cur, kv0, std::false_type{});
        if (more) wait_vmcnt<(NS - 2) * LOADS>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur = (cur + 1 == NS) ? 0 : cur + 1;
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }

    if (prof_on) pt2 = prof_now();
    // ---- epilogue: O[q][h*64 + df*16 + fg*4 + r] = o / l
    bf16_t* Ob = p.O + (int64_t)b * p.strideO + h * 64;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int q = q0 + qi * 16 + fr;
        if (q >= p.Sq) continue;
        const float inv = 1.0f / lacc[qi][0];        // every row of the ones-MFMA result holds the full key sum
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

__global__ void __launch_bounds__(256, 1) attn_small_kernel(const AttnParams p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, blockIdx.x == 0, p.prof_detail);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = bid / p.nq, qt = bid - bh * p.nq;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * (4 * SQW) + w * SQW;
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
    if (prof_on) pt1 = prof_now();

#pragma unroll 1
    for (int blk = 0; blk < SQW / 32; ++blk) {
        const int qb = q0 + blk * 32;
        if (qb >= p.Sq) break;
        frag_ab qf[2][2];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            int q = qb + qi * 16 + fr; if (q > p.Sq - 1) q = p.Sq - 1;
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                const frag_ab raw = *(const frag_ab*)(Qb + (int64_t)q * p.ldq + ds * 32 + fg * 8);
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
    }
    if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt1);
}

}  // namespace

extern "C" int tmix_attn_fwd(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                             const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                             int B, int H, int Sq, int Skv, float scale, void* stream) {
    if (!Q || !K || !Vt || !O) TMIX_FAIL(TMIX_EINVAL, "attn: null pointer");
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) TMIX_FAIL(TMIX_ESHAPE, "attn: empty problem B=%d H=%d Sq=%d Skv=%d", B, H, Sq, Skv);
    if ((int64_t)Skv * ldk >= (1ll << 31) || 64 * ldvt >= (1ll << 31)) TMIX_FAIL(TMIX_ESHAPE, "attn: per-head K/V extent exceeds 32-bit offsets");
    if ((ldq % 8) || (ldk % 8) || (ldvt % 8) || (ldo % 4) || (strideQ % 8) || (strideK % 8) || (strideVt % 8) || (strideO % 4))
        TMIX_FAIL(TMIX_EALIGN, "attn: leading dims / strides must keep 16-byte (Q,K,Vt) / 8-byte (O) alignment");
    if (ldvt < ((Skv + 7) / 8) * 8) TMIX_FAIL(TMIX_ESHAPE, "attn: ldvt=%lld < Skv rounded up to 8", (long long)ldvt);
    if (!aligned16(Q) || !aligned16(K) || !aligned16(Vt) || (((uintptr_t)O) & 7)) TMIX_FAIL(TMIX_EALIGN, "attn: pointer alignment");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_fwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    AttnParams p;
    p.Q = (const bf16_t*)Q; p.ldq = ldq; p.strideQ = strideQ;
    p.K = (const bf16_t*)K; p.ldk = ldk; p.strideK = strideK;
    p.Vt = (const bf16_t*)Vt; p.ldvt = ldvt; p.strideVt = strideVt;
    p.O = (bf16_t*)O; p.ldo = ldo; p.strideO = strideO;
    p.H = H; p.Sq = Sq; p.Skv = Skv; p.nq = (Sq + QB - 1) / QB;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.prof = tmix_prof_take(&p.prof_detail);
    if (Skv <= SK_MAX && !getenv("TMIX_ATTN_GENERAL")) {          // short key set: K / V^T resident in registers, no LDS
        p.nq = (Sq + 4 * SQW - 1) / (4 * SQW);
        const int64_t nws = (int64_t)p.nq * B * H;
        if (nws > 0x7fffffff) TMIX_FAIL(TMIX_ESHAPE, "attn: grid too large");
        attn_small_kernel<<<dim3((unsigned)nws), 256, 0, (hipStream_t)stream>>>(p);
        TMIX_LAUNCH_CHECK();
        return TMIX_OK;
    }
    // the 8-wave form (256-row workgroups) measured SLOWER in situ (S = 1024: 48.2 vs 37.8 us, S = 4096: 267 vs 221): one barrier
    // over eight waves per tile costs more than the halved LDS-DMA issue saves, and two independent 4-wave workgroups per CU drift
    // apart so that one's softmax overlaps the other's MFMAs.  Kept for experiments: TMIX_ATTN_WAVES=8.
    const char* fw = getenv("TMIX_ATTN_WAVES");
    const bool eight = fw && atoi(fw) == 8;
    if (eight) p.nq = (Sq + 255) / 256;
    const int64_t nwg = (int64_t)p.nq * B * H;
    if (nwg > 0x7fffffff) TMIX_FAIL(TMIX_ESHAPE, "attn: grid too large");
    if (eight) attn_fwd_kernel<8><<<dim3((unsigned)nwg), 512, SMEM, (hipStream_t)stream>>>(p);
    else       attn_fwd_kernel<4><<<dim3((unsigned)nwg), 256, SMEM, (hipStream_t)stream>>>(p);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}
