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

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int QB = 128;          // query rows per workgroup (32 per wave)
constexpr int KB = 64;           // keys per tile
constexpr int TILE = KB * 64 * 2;    // 8 KiB (K tile or V^T tile)
constexpr int STAGE = 2 * TILE;
constexpr int SMEM = 2 * STAGE;      // 32 KiB

struct AttnParams {
    const bf16_t* Q; int64_t ldq, strideQ;
    const bf16_t* K; int64_t ldk, strideK;
    const bf16_t* Vt; int64_t ldvt, strideVt;
    bf16_t* O; int64_t ldo, strideO;
    int H, Sq, Skv, nq;
    float scale_log2e;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return *(uint32_t*)&v;
}

__global__ void __launch_bounds__(256, 2) attn_fwd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = bid / p.nq, qt = bid - bh * p.nq;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * QB + w * 32;

    const bf16_t* Qb = p.Q + (int64_t)b * p.strideQ + h * 64;
    const bf16_t* Kb = p.K + (int64_t)b * p.strideK + h * 64;
    const bf16_t* Vb = p.Vt + (int64_t)b * p.strideVt + (int64_t)h * 64 * p.ldvt;

    // ---- Q fragments (B operand of S^T = K Q^T), kept in registers for the whole kernel
    frag_ab qf[2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        int q = q0 + qi * 16 + fr; if (q > p.Sq - 1) q = p.Sq - 1;
#pragma unroll
        for (int ds = 0; ds < 2; ++ds)
            qf[qi][ds] = *(const frag_ab*)(Qb + (int64_t)q * p.ldq + ds * 32 + fg * 8);
    }

    // ---- staging geometry: per wave 2 rounds x (8 rows x 8 chunks) for K and for V^T
    const int lrow = lane >> 3;
    const int schunk = ((lane & 7) ^ lrow) * 8;
    int krow[2];                       // key (within tile) whose row lands in this lane's LDS row
    int vrow[2];                       // d row of V^T
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int rho = (r * 4 + w) * 8 + lrow;            // LDS row 0..63
        const int f = rho >> 4, i = rho & 15;
        krow[r] = 32 * (f >> 1) + 8 * (i >> 2) + 4 * (f & 1) + (i & 3);
        vrow[r] = rho;
    }
    const int nt = (p.Skv + KB - 1) / KB;
    auto stage = [&](int buf, int t) {
        char* sK = smem + buf * STAGE;
        char* sV = sK + TILE;
        const int kv0 = t * KB;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int off = (r * 4 + w) * 1024;
            int key = kv0 + krow[r]; if (key > p.Skv - 1) key = p.Skv - 1;
            glds16(Kb + (int64_t)key * p.ldk + schunk, sK + off);
            int64_t c = kv0 + schunk; if (c > p.ldvt - 8) c = p.ldvt - 8;   // fully masked chunk: any finite data
            glds16(Vb + (int64_t)vrow[r] * p.ldvt + c, sV + off);
        }
    };

    f32x4 o[4][2];                     // O^T accumulators: [d fragment][q fragment]
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};

    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) stage(cur ^ 1, t + 1);
        const char* sK = smem + cur * STAGE;
        const char* sV = sK + TILE;

        // ---- S^T fragments: s[f][qi] holds keys kappa(f, 4fg+r), query qi*16+fr
        f32x4 s[4][2];
#pragma unroll
        for (int f = 0; f < 4; ++f) { s[f][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; s[f][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
            const int sw = ((ds * 4 + fg) ^ (fr & 7)) << 4;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const frag_ab kf = *(const frag_ab*)(sK + (f * 16 + fr) * 128 + sw);
                s[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ds], s[f][0], 0, 0, 0);
                s[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ds], s[f][1], 0, 0, 0);
            }
        }

        // ---- online softmax (per query column; keys are spread over r, f and the 4 lane groups)
        const int kv0 = t * KB;
        const bool partial = (kv0 + KB > p.Skv);
        uint32_t pb[2][2][4];          // [qi][k-step p] packed bf16x8 = B operand of O^T = V^T P^T
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            float mx = -INFINITY;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = s[f][qi][r] * p.scale_log2e;
                    if (partial) {
                        const int key = kv0 + 32 * (f >> 1) + 8 * fg + 4 * (f & 1) + r;
                        if (key >= p.Skv) v = -INFINITY;
                    }
                    s[f][qi][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[qi], mx);          // finite: tile 0 always has a valid key
            const float alpha = exp2f(mrun[qi] - mnew);
            mrun[qi] = mnew;
            float sum = 0.f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(s[f][qi][r] - mnew);
                    s[f][qi][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            lrun[qi] = lrun[qi] * alpha + sum;
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[df][qi][r] *= alpha;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                pb[qi][ps][0] = pk_bf16(s[2 * ps][qi][0], s[2 * ps][qi][1]);
                pb[qi][ps][1] = pk_bf16(s[2 * ps][qi][2], s[2 * ps][qi][3]);
                pb[qi][ps][2] = pk_bf16(s[2 * ps + 1][qi][0], s[2 * ps + 1][qi][1]);
                pb[qi][ps][3] = pk_bf16(s[2 * ps + 1][qi][2], s[2 * ps + 1][qi][3]);
            }
        }

        // ---- O^T += V^T P^T
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int sw = ((ps * 4 + fg) ^ (fr & 7)) << 4;
            frag_ab p0, p1;
            __builtin_memcpy(&p0, pb[0][ps], 16);
            __builtin_memcpy(&p1, pb[1][ps], 16);
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                const frag_ab vf = *(const frag_ab*)(sV + (df * 16 + fr) * 128 + sw);
                o[df][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, p0, o[df][0], 0, 0, 0);
                o[df][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, p1, o[df][1], 0, 0, 0);
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: O[q][h*64 + df*16 + fg*4 + r] = o / l
    bf16_t* Ob = p.O + (int64_t)b * p.strideO + h * 64;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int q = q0 + qi * 16 + fr;
        if (q >= p.Sq) continue;
        const float inv = 1.0f / lrun[qi];
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            uint2 v;
            v.x = pk_bf16(o[df][qi][0] * inv, o[df][qi][1] * inv);
            v.y = pk_bf16(o[df][qi][2] * inv, o[df][qi][3] * inv);
            *(uint2*)(Ob + (int64_t)q * p.ldo + df * 16 + fg * 4) = v;
        }
    }
}

}  // namespace

extern "C" int tmix_attn_fwd(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                             const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                             int B, int H, int Sq, int Skv, float scale, void* stream) {
    if (!Q || !K || !Vt || !O) TMIX_FAIL(TMIX_EINVAL, "attn: null pointer");
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) TMIX_FAIL(TMIX_ESHAPE, "attn: empty problem B=%d H=%d Sq=%d Skv=%d", B, H, Sq, Skv);
    if ((ldq % 8) || (ldk % 8) || (ldvt % 8) || (ldo % 4) || (strideQ % 8) || (strideK % 8) || (strideVt % 8) || (strideO % 4))
        TMIX_FAIL(TMIX_EALIGN, "attn: leading dims / strides must keep 16-byte (Q,K,Vt) / 8-byte (O) alignment");
    if (ldvt < ((Skv + 7) / 8) * 8) TMIX_FAIL(TMIX_ESHAPE, "attn: ldvt=%lld < Skv rounded up to 8", (long long)ldvt);
    if (!aligned16(Q) || !aligned16(K) || !aligned16(Vt) || (((uintptr_t)O) & 7)) TMIX_FAIL(TMIX_EALIGN, "attn: pointer alignment");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
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
    const int64_t nwg = (int64_t)p.nq * B * H;
    if (nwg > 0x7fffffff) TMIX_FAIL(TMIX_ESHAPE, "attn: grid too large");
    attn_fwd_kernel<<<dim3((unsigned)nwg), 256, SMEM, (hipStream_t)stream>>>(p);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}
