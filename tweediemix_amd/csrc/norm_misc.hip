// norm_misc.hip -- HBM-bound helpers of the UNet forward on gfx950: GroupNorm(+SiLU) on NHWC,
// LayerNorm, channel concat, sinusoidal timestep embedding, the tiny time/add-embedding MLPs,
// conv_in (4 -> 320, fp32 VALU) and conv_out (320 -> 4, MFMA with fp32 NCHW output).
// All bf16 traffic is 16 bytes per lane.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = bf2f((bf16_t)(v.x & 0xffff)); f[1] = bf2f((bf16_t)(v.x >> 16));
    f[2] = bf2f((bf16_t)(v.y & 0xffff)); f[3] = bf2f((bf16_t)(v.y >> 16));
    f[4] = bf2f((bf16_t)(v.z & 0xffff)); f[5] = bf2f((bf16_t)(v.z >> 16));
    f[6] = bf2f((bf16_t)(v.w & 0xffff)); f[7] = bf2f((bf16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]); v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}

// ------------------------------------------------------------------------------ GroupNorm
constexpr int GN_MAX_C = 4096;

#ifndef GN_T
#define GN_T 128
#endif
#ifndef GN_APPLY_U
#define GN_APPLY_U 2
#endif
#ifndef GN_APPLY_ITEMS
#define GN_APPLY_ITEMS 1024
#endif
#ifndef GN_APPLY_MAXB
#define GN_APPLY_MAXB 4096
#endif
// statistics workgroups per image: up to GN_T, >= 8 pixels each.  A function of the image size ONLY, so the summation order
// -- and with it every output bit -- does not depend on how many images share the launch (co-batched seeds and row-split
// chains reproduce single runs exactly).  Measured (tools/gn_time.py): 32x32 maps want all 128 (18.1 -> 15.1 us against the
// former HW/32 rule); a 32-image video batch would prefer 16-32 per image (-17 %) but that would tie the result to the batch.
__host__ __device__ inline int gn_chunks(int64_t HW) {
    int64_t c = HW / 8;
    if (c > GN_T) c = GN_T;
    if (c < 1) c = 1;
    return (int)c;
}

// partial sums per (batch, chunk of pixels, group): ws[((b*chunks + ch)*groups + g)*2 + {0,1}]
__global__ void __launch_bounds__(256) gn_stats_kernel(const bf16_t* __restrict__ X1, int C1, const bf16_t* __restrict__ X2, int C2,
                                                       float* __restrict__ ws, int64_t HW, int groups, int chunks,
                                                       unsigned long long* prof) {
    __shared__ float s_sum[GN_MAX_C], s_sq[GN_MAX_C];
    if (prof && threadIdx.x == 0) prof_enter(prof, (blockIdx.x | blockIdx.y) == 0, 0);   // in-situ timing (common.h): the norm's three launches share a slot
    const int C = C1 + C2, nvec = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int64_t ppc = (HW + chunks - 1) / chunks;
    const int64_t p0 = (int64_t)ch * ppc;
    int64_t p1 = p0 + ppc; if (p1 > HW) p1 = HW;
    // fixed vector column per thread so the 8 per-channel sums stay in registers across pixels;
    // partials land in LDS at [pixel lane][channel] and are reduced in a fixed order (deterministic).
    const int plane = nvec <= 256 ? 256 / nvec : 1;          // pixel lanes per block; plane * C <= 2048
    for (int v0 = 0; v0 < nvec; v0 += 256) {
        const int v = v0 + (nvec <= 256 ? tid % nvec : tid);
        const int pl = nvec <= 256 ? tid / nvec : 0;
        if (v < nvec && pl < plane) {
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int c0 = v * 8;
            const bf16_t* src; int cs, cl;
            if (c0 < C1) { src = X1 + (int64_t)b * HW * C1; cs = C1; cl = c0; }
            else         { src = X2 + (int64_t)b * HW * C2; cs = C2; cl = c0 - C1; }
            int64_t p = p0 + pl;
            for (; p + 3 * plane < p1; p += 4 * plane) {          // four rows in flight per thread (the loop is latency-bound)
                uint4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) raw[u] = *(const uint4*)(src + (p + u * plane) * cs + cl);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8]; unpack8(raw[u], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { a[j] += f[j]; q[j] += f[j] * f[j]; }
                }
            }
            for (; p < p1; p += plane) {
                const uint4 raw = *(const uint4*)(src + p * cs + cl);
                float f[8]; unpack8(raw, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { a[j] += f[j]; q[j] += f[j] * f[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { s_sum[pl * C + c0 + j] = a[j]; s_sq[pl * C + c0 + j] = q[j]; }
        }
    }
    __syncthreads();
    if (tid < groups) {
        float s = 0.f, q = 0.f;
        for (int pl = 0; pl < plane; ++pl)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { s += s_sum[pl * C + c]; q += s_sq[pl * C + c]; }
        float* o = ws + (((int64_t)b * chunks + ch) * groups + tid) * 2;
        o[0] = s; o[1] = q;
    }
}

// combine the per-chunk partials (fp64) and fold gamma/beta: ss[b][c] = {scale, shift} with
// y = x*scale + shift.  One tiny launch instead of redoing this in every apply workgroup.
__global__ void __launch_bounds__(64) gn_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float2* __restrict__ ss,
                                                         int C, int64_t HW, int groups, int chunks, float eps) {
    // one wave per (group, batch): lanes split the chunks, fp64 tree-combine, then write the group's channels
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, cpg = C / groups;
    double s = 0.0, q = 0.0;
    for (int ch = lane; ch < chunks; ch += 64) {
        const float* o = ws + (((int64_t)b * chunks + ch) * groups + g) * 2;
        s += (double)o[0]; q += (double)o[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean; if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
        const float sc = rstd * gamma[c];
        ss[(int64_t)b * C + c] = make_float2(sc, beta[c] - meanf * sc);
    }
}

// the same from the column partials the tensor's PRODUCERS wrote (tmix_gemm_desc.col_stats_out: [B*HW/32][2][Cs] per source): one
// workgroup of 16 waves per (group, image) adds the group's channels over the image's 32-row blocks -- thread (j, c) walks blocks j, j + J, ...
// of channel c (a fixed order; up to 512 blocks x 80 channels x 2 planes at the 128 x 128 level, where four waves were latency-bound:
// 80 us for the whole norm against 67 with the statistics kernel) in fp32, threads combine in fp64 -- and folds gamma / beta.
// No pass over X: the statistics launch of the three is gone.
constexpr int GN_CS_T = 1024;
__device__ __forceinline__ void gn_cs_walk(const float* __restrict__ base, int Cs, int n, int nblk, int tid, float& s, float& q) {
    if (n <= 0) return;
    const int J = GN_CS_T / n, j = tid / n, c = tid - j * n;       // J >= 4: n <= 256
    if (j >= J) return;
    const float* row = base + (int64_t)j * 2 * Cs + c;
    const int64_t step = (int64_t)J * 2 * Cs;
    int blk = j;
    for (; blk + 3 * J < nblk; blk += 4 * J, row += 4 * step) {     // four blocks in flight per thread
        const float a0 = row[0], b0 = row[Cs], a1 = row[step], b1 = row[step + Cs];
        const float a2 = row[2 * step], b2 = row[2 * step + Cs], a3 = row[3 * step], b3 = row[3 * step + Cs];
        s += (a0 + a1) + (a2 + a3); q += (b0 + b1) + (b2 + b3);
    }
    for (; blk < nblk; blk += J, row += step) { s += row[0]; q += row[Cs]; }
}
__global__ void __launch_bounds__(GN_CS_T) gn_finalize_cs_kernel(const float* __restrict__ cs1, int C1, const float* __restrict__ cs2, int C2,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float2* __restrict__ ss, int64_t HW, int groups, float eps,
                                                                 unsigned long long* prof) {
    __shared__ double s_red[2 * GN_CS_T / 64];
    __shared__ float s_ms[2];
    if (prof && threadIdx.x == 0) prof_enter(prof, (blockIdx.x | blockIdx.y) == 0, 0);
    const int C = C1 + C2, cpg = C / groups;
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int nblk = (int)(HW / 32);
    // gamma / beta of the channel this thread will write: requested before the reduction, whose result they do not depend on
    float gm = 0.f, bt = 0.f;
    if (tid < cpg) { gm = gamma[g * cpg + tid]; bt = beta[g * cpg + tid]; }
    // the group's channels that live in source 1 / source 2 (a group may straddle the two)
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    const int n1 = min(c_hi, C1) - min(c_lo, C1), n2 = cpg - n1;
    float s = 0.f, q = 0.f;
    gn_cs_walk(cs1 + (int64_t)b * nblk * 2 * C1 + c_lo, C1, n1, nblk, tid, s, q);
    if (n2 > 0) gn_cs_walk(cs2 + (int64_t)b * nblk * 2 * C2 + (max(c_lo, C1) - C1), C2, n2, nblk, tid, s, q);
    double sd = (double)s, qd = (double)q;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sd += __shfl_xor(sd, o); qd += __shfl_xor(qd, o); }
    if ((tid & 63) == 0) { s_red[(tid >> 6) * 2] = sd; s_red[(tid >> 6) * 2 + 1] = qd; }
    __syncthreads();
    if (tid == 0) {
        double st = 0.0, qt = 0.0;
#pragma unroll
        for (int w = 0; w < GN_CS_T / 64; ++w) { st += s_red[2 * w]; qt += s_red[2 * w + 1]; }
        const double n = (double)HW * cpg;
        const double mean = st / n;
        double var = qt / n - mean * mean; if (var < 0.0) var = 0.0;
        s_ms[0] = (float)mean; s_ms[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    if (tid < cpg) {
        const float sc = s_ms[1] * gm;
        ss[(int64_t)b * C + c_lo + tid] = make_float2(sc, bt - s_ms[0] * sc);
    }
}

__device__ __forceinline__ float silu_fast(float x) {     // x * sigmoid(x) with v_exp_f32 / v_rcp_f32
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// maximum over the four lanes of a quad (DPP quad_perm moves): the MX block of the e4m3 form is the 4 adjacent 8-channel vectors of a pixel
__device__ __forceinline__ float gn_quad_max(float x) {
    float y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));
    x = fmaxf(x, y);
    y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));
    return fmaxf(x, y);
}

// F8 = 1 (tmix_groupnorm_nhwc_pre_f8): the normalised (+ SiLU) tensor leaves as OCP e4m3 bytes [B*HW][C] with one E8M0 scale per (pixel, 32 channels) in the
// ROW-major form [B*HW][C / 32] -- the input of tmix_conv3x3_nhwc_fp8 (a tap shift moves a pixel's scales by a multiple of 4 bytes) -- exactly what an MX
// quantiser makes of the bf16 tensor the plain kernel writes.  Work items are dealt out in multiples of four vectors so that a quad of lanes holds one block.
template <int F8>
__global__ void __launch_bounds__(256) gn_apply_kernel(const bf16_t* __restrict__ X1, int C1, const bf16_t* __restrict__ X2, int C2,
                                                       bf16_t* __restrict__ Y, const float2* __restrict__ ss, int64_t HW, int silu,
                                                       unsigned long long* prof, unsigned char* __restrict__ Y8 = nullptr, unsigned char* __restrict__ S8 = nullptr) {
    __shared__ float2 s_ss[GN_MAX_C];
    const unsigned long long pt0 = (prof && threadIdx.x == 0) ? prof_now() : 0;
    const int C = C1 + C2, nvec = C >> 3;
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) s_ss[c] = ss[(int64_t)b * C + c];
    __syncthreads();
    const int64_t total = HW * nvec;
    int64_t per = (total + gridDim.x - 1) / gridDim.x;
    if (F8) per = (per + 3) & ~(int64_t)3;
    const int64_t i0 = (int64_t)blockIdx.x * per;
    int64_t i1 = i0 + per; if (i1 > total) i1 = total;
    int64_t p = i0 / nvec; int v = (int)(i0 - p * nvec) + tid;      // running (pixel, vector) cursor: no 64-bit division per item
    while (v >= nvec) { v -= nvec; ++p; }
    const int step_p = 256 / nvec, step_v = 256 - step_p * nvec;
    constexpr int U = GN_APPLY_U;                        // loads in flight per thread (the pass is latency-bound)
    for (int64_t i = i0 + tid; i < i1; i += 256 * U) {
        uint4 raw[U]; int64_t pp[U]; int cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pp[u] = p; cc[u] = v * 8;
            if (i + u * 256 < i1) {
                const bf16_t* src = (cc[u] < C1) ? X1 + ((int64_t)b * HW + p) * C1 + cc[u] : X2 + ((int64_t)b * HW + p) * C2 + (cc[u] - C1);
                raw[u] = *(const uint4*)src;
            }
            p += step_p; v += step_v;
            if (v >= nvec) { v -= nvec; ++p; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i + u * 256 < i1) {
                float f[8]; unpack8(raw[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 k = s_ss[cc[u] + j];
                    float y = f[j] * k.x + k.y;
                    if (silu) y = silu_fast(y);
                    f[j] = y;
                }
                if constexpr (F8) {
                    const uint4 pk = pack8(f);
                    unpack8(pk, f);                       // the bf16-rounded values
                    float am = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(f[j]));
                    am = gn_quad_max(am);
                    const int e = e8m0_for_amax(am);
                    const float inv = exp2_neg_int(e);
                    int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false); q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, q0, true);
                    int q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, q1, true);
                    const int64_t row = (int64_t)b * HW + pp[u];
                    *(uint2*)(Y8 + row * C + cc[u]) = make_uint2((unsigned)q0, (unsigned)q1);
                    if ((tid & 3) == 0) S8[row * (C >> 5) + (cc[u] >> 5)] = (unsigned char)(e + 127);
                } else
                *(uint4*)(Y + ((int64_t)b * HW + pp[u]) * C + cc[u]) = pack8(f);
            }
        }
    }
    if (prof && threadIdx.x == 0) prof_leave(prof, 0, pt0, pt0, pt0);
}

// GroupNorm (+ SiLU) of a SMALL image in ONE launch (tmix_groupnorm_nhwc picks it by shape): one workgroup owns gpw consecutive groups of one image -- cw = gpw * C / groups
// channels, a multiple of 8 -- reads its HW x cw slice twice (statistics, then apply: <= 128 KB, it stays in L2) and needs nobody else's sums.  The three-launch form costs a
// 336-pixel frame of the video UNet's third level 10 + 5 + 10 us and two kernel boundaries; the slices here are 27 - 80 KB.  Thread t keeps vector t % nv of rows t / nv,
// + rs, ...; the sums meet in LDS and are added in a fixed order (channel sums over row slots, then group sums over channels, both in fp64): every output bit is a function
// of the image alone, as with the other forms.
constexpr int GN_SMALL_MAX_CW = 256, GN_SMALL_MAX_ELEMS = 65536;
__global__ void __launch_bounds__(256) gn_small_kernel(const bf16_t* __restrict__ X1, int C1, const bf16_t* __restrict__ X2, int C2, bf16_t* __restrict__ Y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int groups, int gpw, float eps, int silu,
                                                       unsigned long long* prof) {
    __shared__ float r_s[2048], r_q[2048];
    __shared__ double c_s[GN_SMALL_MAX_CW], c_q[GN_SMALL_MAX_CW];
    __shared__ float g_ms[16];
    __shared__ float2 s_ss[GN_SMALL_MAX_CW];
    const unsigned long long pt0 = (prof && threadIdx.x == 0) ? prof_enter(prof, (blockIdx.x | blockIdx.y) == 0, 0) : 0;      // (the one launch writes both stamps of its slot)
    const int C = C1 + C2, cpg = C / groups, cw = gpw * cpg, nv = cw >> 3, rs = 256 / nv;
    const int b = blockIdx.y, c0 = blockIdx.x * cw, tid = threadIdx.x;
    const int v = tid % nv, slot = tid / nv;
    const bool active = slot < rs;
    const int cg = c0 + v * 8;
    const bf16_t* src; int ld;
    if (cg < C1) { src = X1 + (int64_t)b * HW * C1 + cg; ld = C1; } else { src = X2 + (int64_t)b * HW * C2 + (cg - C1); ld = C2; }
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (active) {
        int row = slot;
        for (; row + 3 * rs < HW; row += 4 * rs) {                 // four rows in flight
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *(const uint4*)(src + (int64_t)(row + u * rs) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8]; unpack8(raw[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
            }
        }
        for (; row < HW; row += rs) {
            float f[8]; unpack8(*(const uint4*)(src + (int64_t)row * ld), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { r_s[slot * cw + v * 8 + j] = s[j]; r_q[slot * cw + v * 8 + j] = q[j]; }
    }
    __syncthreads();
    if (tid < cw) {
        double a = 0.0, d = 0.0;
        for (int k = 0; k < rs; ++k) { a += (double)r_s[k * cw + tid]; d += (double)r_q[k * cw + tid]; }
        c_s[tid] = a; c_q[tid] = d;
    }
    __syncthreads();
    if (tid < gpw) {
        double a = 0.0, d = 0.0;
        for (int k = 0; k < cpg; ++k) { a += c_s[tid * cpg + k]; d += c_q[tid * cpg + k]; }
        const double n = (double)HW * cpg, mean = a / n;
        double var = d / n - mean * mean; if (var < 0.0) var = 0.0;
        g_ms[2 * tid] = (float)mean; g_ms[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    if (tid < cw) {
        const int g = tid / cpg;
        const float sc = g_ms[2 * g + 1] * gamma[c0 + tid];
        s_ss[tid] = make_float2(sc, beta[c0 + tid] - g_ms[2 * g] * sc);
    }
    __syncthreads();
    if (active) {
        float2 k[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = s_ss[v * 8 + j];
        bf16_t* dst = Y + (int64_t)b * HW * C + cg;
        int row = slot;
        for (; row + 3 * rs < HW; row += 4 * rs) {
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *(const uint4*)(src + (int64_t)(row + u * rs) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8]; unpack8(raw[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float y = f[j] * k[j].x + k[j].y; if (silu) y = silu_fast(y); f[j] = y; }
                *(uint4*)(dst + (int64_t)(row + u * rs) * C) = pack8(f);
            }
        }
        for (; row < HW; row += rs) {
            float f[8]; unpack8(*(const uint4*)(src + (int64_t)row * ld), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { float y = f[j] * k[j].x + k[j].y; if (silu) y = silu_fast(y); f[j] = y; }
            *(uint4*)(dst + (int64_t)row * C) = pack8(f);
        }
    }
    if (prof && threadIdx.x == 0) prof_leave(prof, 0, pt0, pt0, pt0);
}
// gpw for the one-launch form, or 0 when the shape does not qualify (the slice must be small).  A function of the image's shape ONLY -- not of the batch: co-batched seeds
// and row-split chains must take the same path as a single run to reproduce it bit for bit.
static int gn_small_gpw(int64_t HW, int C, int groups) {
    if (tmix_env(TMIX_ENV_GN_NO_SMALL)) return 0;
    const int cpg = C / groups;
    for (int gpw = 1; gpw <= 8; gpw <<= 1) {
        if (groups % gpw || ((gpw * cpg) & 7)) continue;
        const int cw = gpw * cpg;
        if (cw > GN_SMALL_MAX_CW || HW * cw > GN_SMALL_MAX_ELEMS) return 0;
        return gpw;
    }
    return 0;
}

// ------------------------------------------------------------------------------ LayerNorm
// one wave per row, row kept in registers (exact two-pass variance); C <= 2048, C % 8 == 0
__global__ void __launch_bounds__(256) layernorm_kernel(const bf16_t* __restrict__ X, bf16_t* __restrict__ Y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = C >> 3;
    float f[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = i * 64 + lane;
        if (v < nvec) {
            const uint4 raw = *(const uint4*)(X + row * C + v * 8);
            unpack8(raw, f[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[i][j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = i * 64 + lane;
        if (v < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = i * 64 + lane;
        if (v < nvec) {
            const float4 g0 = *(const float4*)(gamma + v * 8), g1 = *(const float4*)(gamma + v * 8 + 4);
            const float4 b0 = *(const float4*)(beta + v * 8), b1 = *(const float4*)(beta + v * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (f[i][j] - mean) * rstd * gg[j] + bb[j];
            *(uint4*)(Y + row * C + v * 8) = pack8(y);
        }
    }
}

// ------------------------------------------------------------------------------ row softmax (fp32 -> bf16)
// one workgroup per row; the row (<= 64K columns) is streamed three times from L2/HBM (max, sum, write).
// seq > 0: causal rows of a [.., seq, cols] score stack -- row r attends to columns <= r % seq, the rest get P = 0.
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ S, int64_t ld_s, bf16_t* __restrict__ P,
                                                           int64_t ld_p, int cols, float scale_log2e, int seq, int valid) {
    __shared__ float red[4];
    const float* row = S + (int64_t)blockIdx.x * ld_s;
    bf16_t* out = P + (int64_t)blockIdx.x * ld_p;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lim = seq > 0 ? (int)(blockIdx.x % (unsigned)seq) + 1 : valid;    // visible columns: [0, lim)
    auto load = [&](int c) {
        float4 v = *(const float4*)(row + c);
        if (c + 0 >= lim) v.x = -INFINITY;
        if (c + 1 >= lim) v.y = -INFINITY;
        if (c + 2 >= lim) v.z = -INFINITY;
        if (c + 3 >= lim) v.w = -INFINITY;
        return v;
    };
    float mx = -INFINITY;
    for (int c = tid * 4; c < cols; c += 1024) {
        const float4 v = load(c);
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2e;
    __syncthreads();
    float sum = 0.f;
    for (int c = tid * 4; c < cols; c += 1024) {
        const float4 v = load(c);
        sum += exp2f(v.x * scale_log2e - mx) + exp2f(v.y * scale_log2e - mx) + exp2f(v.z * scale_log2e - mx) + exp2f(v.w * scale_log2e - mx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[w] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = tid * 4; c < cols; c += 1024) {
        const float4 v = load(c);
        uint2 o;
        o.x = pack_bf2(exp2f(v.x * scale_log2e - mx) * inv, exp2f(v.y * scale_log2e - mx) * inv);
        o.y = pack_bf2(exp2f(v.z * scale_log2e - mx) * inv, exp2f(v.w * scale_log2e - mx) * inv);
        *(uint2*)(out + c) = o;
    }
}

__global__ void __launch_bounds__(256) affine_clamp_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                           float scale, float shift, float lo, float hi) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        y[i] = fminf(fmaxf(x[i] * scale + shift, lo), hi);
}

// ------------------------------------------------------------------------------ concat
__global__ void __launch_bounds__(256) concat_kernel(const uint4* __restrict__ X1, int v1, const uint4* __restrict__ X2, int v2,
                                                     uint4* __restrict__ Y, int64_t rows) {
    const int nv = v1 + v2;
    const int64_t total = rows * nv;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / nv; const int v = (int)(i - r * nv);
        Y[i] = v < v1 ? X1[r * v1 + v] : X2[r * v2 + (v - v1)];
    }
}

// ------------------------------------------------------------------------------ timestep embedding
__global__ void timestep_embedding_kernel(const float* __restrict__ values, float* __restrict__ out, int count, int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * half) return;
    const int r = i / half, j = i - r * half;
    const float freq = expf(-9.210340371976184f * (float)j / (float)half);     // ln(10000)
    const float a = values[r] * freq;
    out[(int64_t)r * dim + j] = cosf(a);                 // flip_sin_to_cos=True: [cos | sin]
    out[(int64_t)r * dim + half + j] = sinf(a);
}

// ------------------------------------------------------------------------------ small-M linear
// one wave per output column; the M (<=16) input rows are tiny and L1/L2 resident
template <int MAXM>
__global__ void __launch_bounds__(256) linear_small_kernel(const float* __restrict__ in, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ bias, const float* __restrict__ add,
                                                           float* __restrict__ out, int M, int N, int K, int act_in, int act_out,
                                                           const int* __restrict__ secs, int nsec, int Mtot, int mrow0) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) acc[m] = 0.f;
    const int nvec = K >> 3;
    for (int v = lane; v < nvec; v += 64) {
        const uint4 raw = *(const uint4*)(W + (int64_t)n * K + v * 8);
        float wf[8]; unpack8(raw, wf);
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (m < M) {
                const float4 x0 = *(const float4*)(in + (int64_t)m * K + v * 8);
                const float4 x1 = *(const float4*)(in + (int64_t)m * K + v * 8 + 4);
                float xf[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                if (act_in) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[j] = silu_f(xf[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[m] += xf[j] * wf[j];
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[m] += __shfl_xor(acc[m], o);
    }
    if (lane == 0) {
        // sections: columns [secs[s], secs[s+1]) leave as their own dense [M][width] matrix at out + secs[s] * M
        int64_t base = 0; int ldn = N, col = n;
        if (secs) {
            int sidx = 0;
            while (sidx + 1 < nsec && secs[sidx + 1] <= n) ++sidx;
            base = (int64_t)secs[sidx] * Mtot; ldn = secs[sidx + 1] - secs[sidx]; col = n - secs[sidx];
        }
        for (int m = 0; m < M; ++m) {
            float y = acc[m] + (bias ? bias[n] : 0.f) + (add ? add[(int64_t)m * N + n] : 0.f);
            if (act_out) y = silu_f(y);
            out[base + (int64_t)(secs ? mrow0 + m : m) * ldn + col] = y;
        }
    }
}

// ------------------------------------------------------------------------------ conv_in (fp32 VALU)
// workgroup = 64 consecutive output pixels x all Cout; lane = pixel (its 3x3xCIN patch lives in registers), wave q
// owns a quarter of the output channels; the fp32 OHWI weights are staged once per workgroup in LDS and read as
// wave-uniform broadcasts.  fp32 math (the latent is not rounded to bf16 before the first convolution).
struct PreMap { float w[16]; float b[4]; int on; };

template <int CIN>
__global__ void __launch_bounds__(256) conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                      int B, int H, int W, int Cout, const PreMap pm) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];        // [Cout][9*CIN]
    constexpr int KK = 9 * CIN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < Cout * KK / 4; i += 256) ((float4*)s_w)[i] = ((const float4*)w)[i];
    __syncthreads();
    const int64_t npix = (int64_t)B * H * W;
    // persistent over 64-pixel tiles: the weight stage (up to 92 KB) is paid once per workgroup, not once per 64 pixels
    for (int64_t tile = blockIdx.x; tile * 64 < npix; tile += gridDim.x) {
    int64_t pix = tile * 64 + lane;
    const bool live = pix < npix;
    if (!live) pix = npix - 1;
    const int b = (int)(pix / ((int64_t)H * W)); const int rem = (int)(pix - (int64_t)b * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    float patch[KK];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            float v[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) v[ci] = ok ? x[(((int64_t)b * CIN + ci) * H + iy) * W + ix] : 0.f;
            if (pm.on && CIN == 4) {          // per-pixel linear map of the latent; padding stays exactly zero
                float u[4];
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    u[co] = ok ? pm.b[co] + pm.w[co * 4 + 0] * v[0] + pm.w[co * 4 + 1] * v[1] + pm.w[co * 4 + 2] * v[2] + pm.w[co * 4 + 3] * v[3] : 0.f;
#pragma unroll
                for (int co = 0; co < 4; ++co) v[co] = u[co];
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) patch[(ky * 3 + kx) * CIN + ci] = v[ci];
        }
    const int cpq = Cout / 4;                      // output channels per wave (multiple of 8)
    for (int c0 = q * cpq; c0 < (q + 1) * cpq; c0 += 8) {
        float o[8];
        // (CIN = 8: 72-tap patches; unrolling all 8 filters at once spills 160 registers to scratch and runs 6x slower per flop)
#pragma unroll 2
        for (int j = 0; j < 8; ++j) {
            const float* wr = s_w + (c0 + j) * KK;
            float a = bias ? bias[c0 + j] : 0.f;
#pragma unroll
            for (int k = 0; k < KK; ++k) a += patch[k] * wr[k];
            o[j] = a;
        }
        if (live) *(uint4*)(y + pix * Cout + c0) = pack8(o);
    }
    }
}

// ------------------------------------------------------------------------------ conv_out (MFMA)
// wave = 16 output pixels x (<=16 padded) output channels; K = 9*Cin streamed straight from global
// (A fragment = 16 B of one pixel's channels per lane, W fragment = 16 B of one filter row per lane).
__global__ void __launch_bounds__(256) conv_out_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       int B, int H, int W, int Cin, int Cout) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const int64_t npix = (int64_t)B * H * W;
    const int64_t pbase = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (pbase >= npix) return;
    int64_t pix = pbase + fr; if (pix > npix - 1) pix = npix - 1;
    const int b = (int)(pix / ((int64_t)H * W)); const int rem = (int)(pix - (int64_t)b * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    const int co = fr < Cout ? fr : Cout - 1;                 // padded filter rows replicate a real one
    const bf16_t* wrow = w + (int64_t)co * 9 * Cin;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const frag_ab zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const bf16_t* src = x + (((int64_t)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cin;
        for (int c = 0; c < Cin; c += 32) {
            frag_ab a = *(const frag_ab*)(src + c + fg * 8);
            if (!ok) a = zero;
            const frag_ab wf = *(const frag_ab*)(wrow + tap * Cin + c + fg * 8);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf, acc, 0, 0, 0);   // D[pixel][cout]
        }
    }
    // lane holds pixels pbase + 4*fg + r for output channel fr
    if (fr < Cout) {
        const float bv = bias ? bias[fr] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t pp = pbase + fg * 4 + r;
            if (pp < npix) {
                const int bb = (int)(pp / ((int64_t)H * W)); const int64_t rr = pp - (int64_t)bb * H * W;
                y[((int64_t)bb * Cout + fr) * H * W + rr] = acc[r] + bv;
            }
        }
    }
}

}  // namespace

extern "C" int tmix_groupnorm_ws_chunks(int64_t HW) { return gn_chunks(HW); }
extern "C" int64_t tmix_groupnorm_ws_floats(int B, int C, int groups) { return (int64_t)B * GN_T * groups * 2 + (int64_t)B * C * 2; }

extern "C" int tmix_groupnorm_nhwc_launches(int64_t HW, int C, int groups) {
    if (HW <= 0 || C <= 0 || groups <= 0 || (C % groups)) return 0;
    return gn_small_gpw(HW, C, groups) ? 1 : 3;
}

extern "C" int tmix_groupnorm_nhwc(const void* X1, int C1, const void* X2, int C2, void* Y, const float* gamma,
                                   const float* beta, float* ws, int B, int64_t HW, int groups, float eps, int silu,
                                   void* stream) {
    if (!X1 || !Y || !gamma || !beta || !ws) TMIX_FAIL(TMIX_EINVAL, "groupnorm: null pointer");
    if (C2 > 0 && !X2) TMIX_FAIL(TMIX_EINVAL, "groupnorm: C2 > 0 but X2 is null");
    const int C = C1 + C2;
    if (B <= 0 || HW <= 0 || C <= 0) TMIX_FAIL(TMIX_ESHAPE, "groupnorm: empty problem");
    if ((C1 % 8) || (C2 % 8) || C > GN_MAX_C || groups <= 0 || groups > 64 || (C % groups)) TMIX_FAIL(TMIX_ESHAPE, "groupnorm: C1=%d C2=%d groups=%d unsupported", C1, C2, groups);
    if (!aligned16(X1) || (X2 && !aligned16(X2)) || !aligned16(Y)) TMIX_FAIL(TMIX_EALIGN, "groupnorm: pointers must be 16-byte aligned");
    const int chunks = gn_chunks(HW);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* prof = tmix_prof_take();
    if (const int gpw = gn_small_gpw(HW, C, groups)) {          // small images: statistics + apply in one launch, one workgroup per (image, gpw groups)
        gn_small_kernel<<<dim3(groups / gpw, B), 256, 0, st>>>((const bf16_t*)X1, C1, (const bf16_t*)X2, C2, (bf16_t*)Y, gamma, beta, (int)HW, groups, gpw, eps, silu, prof);
        TMIX_LAUNCH_CHECK();
        return TMIX_OK;
    }
    gn_stats_kernel<<<dim3(chunks, B), 256, 0, st>>>((const bf16_t*)X1, C1, (const bf16_t*)X2, C2, ws, HW, groups, chunks, prof);
    TMIX_LAUNCH_CHECK();
    // ws layout: [B*chunks*groups*2] partial sums | [B*C] float2 scale/shift
    float2* ss = (float2*)(ws + (int64_t)B * GN_T * groups * 2);
    gn_finalize_kernel<<<dim3(groups, B), 64, 0, st>>>(ws, gamma, beta, ss, C, HW, groups, chunks, eps);
    TMIX_LAUNCH_CHECK();
    int64_t nb = (HW * (C / 8) + GN_APPLY_ITEMS - 1) / GN_APPLY_ITEMS; if (nb < 1) nb = 1; if (nb > GN_APPLY_MAXB) nb = GN_APPLY_MAXB;
    gn_apply_kernel<0><<<dim3((unsigned)nb, B), 256, 0, st>>>((const bf16_t*)X1, C1, (const bf16_t*)X2, C2, (bf16_t*)Y, ss, HW, silu, prof);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

static int gn_pre_entry(const void* X1, int C1, const void* X2, int C2, void* Y, void* S8, const float* gamma,
                        const float* beta, float* ws, int B, int64_t HW, int groups, float eps, int silu,
                        const float* cs1, int cs1_channels, const float* cs2, int cs2_channels, void* stream) {
    if (!X1 || !Y || !gamma || !beta || !ws || !cs1) TMIX_FAIL(TMIX_EINVAL, "groupnorm_pre: null pointer");
    if (S8 && ((C1 + C2) % 32)) TMIX_FAIL(TMIX_ESHAPE, "groupnorm_pre_f8: C must be a multiple of 32 (MX blocks)");
    if (C2 > 0 && !X2) TMIX_FAIL(TMIX_EINVAL, "groupnorm_pre: C2 > 0 but X2 is null");
    const int C = C1 + C2;
    if (cs1_channels <= 0 || cs2_channels < 0 || cs1_channels + cs2_channels != C || (cs2_channels > 0 && !cs2))
        TMIX_FAIL(TMIX_EINVAL, "groupnorm_pre: the partials cover %d + %d channels, the tensor has %d", cs1_channels, cs2_channels, C);
    if (B <= 0 || HW <= 0 || C <= 0) TMIX_FAIL(TMIX_ESHAPE, "groupnorm_pre: empty problem");
    if ((C1 % 8) || (C2 % 8) || C > GN_MAX_C || groups <= 0 || groups > 64 || (C % groups) || C / groups > 256) TMIX_FAIL(TMIX_ESHAPE, "groupnorm_pre: C1=%d C2=%d groups=%d unsupported", C1, C2, groups);
    if (HW % TMIX_COLSTATS_ROWS) TMIX_FAIL(TMIX_ESHAPE, "groupnorm_pre: HW=%lld must be a multiple of %d (the producers' partials cover 32-row blocks)", (long long)HW, TMIX_COLSTATS_ROWS);
    if (!aligned16(X1) || (X2 && !aligned16(X2)) || !aligned16(Y)) TMIX_FAIL(TMIX_EALIGN, "groupnorm_pre: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* prof = tmix_prof_take();
    float2* ss = (float2*)(ws + (int64_t)B * GN_T * groups * 2);           // the same workspace layout as tmix_groupnorm_nhwc
    gn_finalize_cs_kernel<<<dim3(groups, B), GN_CS_T, 0, st>>>(cs1, cs1_channels, cs2, cs2_channels, gamma, beta, ss, HW, groups, eps, prof);
    TMIX_LAUNCH_CHECK();
    int64_t nb = (HW * (C / 8) + GN_APPLY_ITEMS - 1) / GN_APPLY_ITEMS; if (nb < 1) nb = 1; if (nb > GN_APPLY_MAXB) nb = GN_APPLY_MAXB;
    if (S8) gn_apply_kernel<1><<<dim3((unsigned)nb, B), 256, 0, st>>>((const bf16_t*)X1, C1, (const bf16_t*)X2, C2, nullptr, ss, HW, silu, prof, (unsigned char*)Y, (unsigned char*)S8);
    else gn_apply_kernel<0><<<dim3((unsigned)nb, B), 256, 0, st>>>((const bf16_t*)X1, C1, (const bf16_t*)X2, C2, (bf16_t*)Y, ss, HW, silu, prof);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_groupnorm_nhwc_pre(const void* X1, int C1, const void* X2, int C2, void* Y, const float* gamma,
                                       const float* beta, float* ws, int B, int64_t HW, int groups, float eps, int silu,
                                       const float* cs1, int cs1_channels, const float* cs2, int cs2_channels, void* stream) {
    return gn_pre_entry(X1, C1, X2, C2, Y, nullptr, gamma, beta, ws, B, HW, groups, eps, silu, cs1, cs1_channels, cs2, cs2_channels, stream);
}

extern "C" int tmix_groupnorm_nhwc_pre_f8(const void* X1, int C1, const void* X2, int C2, void* Y8, void* scales, const float* gamma,
                                          const float* beta, float* ws, int B, int64_t HW, int groups, float eps, int silu,
                                          const float* cs1, int cs1_channels, const float* cs2, int cs2_channels, void* stream) {
    if (!scales) TMIX_FAIL(TMIX_EINVAL, "groupnorm_pre_f8: null scale array");
    return gn_pre_entry(X1, C1, X2, C2, Y8, scales, gamma, beta, ws, B, HW, groups, eps, silu, cs1, cs1_channels, cs2, cs2_channels, stream);
}

extern "C" int tmix_layernorm(const void* X, void* Y, const float* gamma, const float* beta, int64_t rows, int C,
                              float eps, void* stream) {
    if (!X || !Y || !gamma || !beta) TMIX_FAIL(TMIX_EINVAL, "layernorm: null pointer");
    if (rows <= 0 || C <= 0) TMIX_FAIL(TMIX_ESHAPE, "layernorm: empty problem");
    if ((C % 8) || C > 2048) TMIX_FAIL(TMIX_ESHAPE, "layernorm: C=%d must be a multiple of 8 and <= 2048", C);
    if (!aligned16(X) || !aligned16(Y) || !aligned16(gamma) || !aligned16(beta)) TMIX_FAIL(TMIX_EALIGN, "layernorm: pointers must be 16-byte aligned");
    layernorm_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>((const bf16_t*)X, (bf16_t*)Y, gamma, beta, rows, C, eps);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

// ------------------------------------------------------------------------------ fp8 row quantiser
// one wave per row: q[r][k] = e4m3(x[r][k] * 2^-(e_r - 127)) with e_r the smallest E8M0 exponent that brings the row's largest
// magnitude under 448 (the e4m3 maximum); an all-zero row gets e = 127 (scale 1).  The row stays in registers between the
// maximum and the conversion (one HBM read, K <= 8192).
__global__ void __launch_bounds__(256) quantize_fp8_rows_kernel(const bf16_t* __restrict__ X, int64_t ld, unsigned char* __restrict__ Q, int64_t ldq,
                                                                unsigned char* __restrict__ scale, int64_t rows, int K) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    constexpr int MAXV = 16;                                  // 16-byte vectors per lane: K <= 64 * 8 * 16 = 8192
    const int nv = K >> 3;
    uint4 v[MAXV];
    float amax = 0.f;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int c = u * 64 + lane;
        if (c < nv) {
            v[u] = *(const uint4*)(X + r * ld + (int64_t)c * 8);
            const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) amax = fmaxf(amax, fmaxf(fabsf(__uint_as_float(w[k] << 16)), fabsf(__uint_as_float(w[k] & 0xffff0000u))));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    const int e = e8m0_for_amax(amax);
    const float inv = exp2_neg_int(e);
    if (lane == 0) scale[r] = (unsigned char)(e + 127);
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int c = u * 64 + lane;
        if (c < nv) {
            const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            unsigned o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float f0 = __uint_as_float(w[2 * h] << 16) * inv, f1 = __uint_as_float(w[2 * h] & 0xffff0000u) * inv;
                const float f2 = __uint_as_float(w[2 * h + 1] << 16) * inv, f3 = __uint_as_float(w[2 * h + 1] & 0xffff0000u) * inv;
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(f2, f3, pk, true);
                o[h] = (unsigned)pk;
            }
            *(uint2*)(Q + r * ldq + (int64_t)c * 8) = make_uint2(o[0], o[1]);
        }
    }
}

extern "C" int tmix_quantize_fp8_rows(const void* X, int64_t ld, void* Q, int64_t ldq, uint8_t* scale_e8m0, int64_t rows, int K, void* stream) {
    if (!X || !Q || !scale_e8m0) TMIX_FAIL(TMIX_EINVAL, "quantize_fp8_rows: null pointer");
    if (rows <= 0 || K <= 0 || (K % 8) || K > 8192) TMIX_FAIL(TMIX_ESHAPE, "quantize_fp8_rows: rows=%lld K=%d (K %% 8 == 0, K <= 8192)", (long long)rows, K);
    if (!aligned16(X) || (ld % 8) || (((uintptr_t)Q) & 7) || (ldq % 8)) TMIX_FAIL(TMIX_EALIGN, "quantize_fp8_rows: X rows must be 16-byte, Q rows 8-byte aligned");
    quantize_fp8_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>((const bf16_t*)X, ld, (unsigned char*)Q, ldq, scale_e8m0, rows, K);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_zero(void* ptr, int64_t nbytes, void* stream) {
    if (!ptr || nbytes <= 0) TMIX_FAIL(TMIX_EINVAL, "zero: null pointer / empty range");
    hipError_t e = hipMemsetAsync(ptr, 0, (size_t)nbytes, (hipStream_t)stream);
    if (e != hipSuccess) TMIX_FAIL((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    return TMIX_OK;
}

extern "C" int tmix_concat_channels(const void* X1, int C1, const void* X2, int C2, void* Y, int64_t rows, void* stream) {
    if (!X1 || !X2 || !Y) TMIX_FAIL(TMIX_EINVAL, "concat: null pointer");
    if (rows <= 0 || C1 <= 0 || C2 <= 0 || (C1 % 8) || (C2 % 8)) TMIX_FAIL(TMIX_ESHAPE, "concat: rows=%lld C1=%d C2=%d unsupported", (long long)rows, C1, C2);
    if (!aligned16(X1) || !aligned16(X2) || !aligned16(Y)) TMIX_FAIL(TMIX_EALIGN, "concat: pointers must be 16-byte aligned");
    int64_t nb = (rows * ((C1 + C2) / 8) + 255) / 256; if (nb > 4096) nb = 4096;
    concat_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>((const uint4*)X1, C1 / 8, (const uint4*)X2, C2 / 8, (uint4*)Y, rows);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

// ------------------------------------------------------------------------------ LoRA down-projection (low-rank form of the concept deltas)
// utils_lora.py:65-79,113-119 add `up(down(x))` of concept i to batch row i + 1 of every attention projection (model_lora.py:41-48,
// rank 4).  In the low-rank mode (UNetWeights(lora_mode="lowrank")) the projection runs ONCE on shared weights [W | U | 0] with
// K + 64 input columns: the 64 pad columns behind a row of A hold that row's down-projections -- the P = 4 x (projections fused in the
// GEMM) values of the row's OWN concept at columns K + set * P .., zeros elsewhere -- so the GEMM's last K-tile adds up(down(x)) and no
// merged per-concept weight copies exist.  This kernel fills the pad.  With a LayerNorm folded into the GEMM (ln != 0) the GEMM forms
// rstd * (acc - mean * colsum(W')) + bias with colsum over the first K columns only, so the pad must hold T / rstd where
// T = LN(x) D^T:  (x - mean) D'^T + (D beta) / rstd  with D' = D * gamma; mean / rstd are taken from the row itself (fp32, E[x^2] - mean^2,
// the same definition the GEMM's statistics use).
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 lora_frag;
// One workgroup = 16 rows of A, its four waves split K: per 32-wide k-step a wave issues v_mfma_f32_16x16x32_bf16 twice --
//   C1[i][j] += sum_k Dx[i][k] X[j][k]   Dx = the concept's P down rows, then one row of ones (row P: the row sum s1), zeros
//   C2[i][j] += sum_k X[i][k] X[j][k]    the Gram matrix of the 16 rows: its diagonal is the sum of squares s2 (exact: bf16 products, fp32 sums)
// -- both operands straight from global memory in MFMA layout (lane = (row, k-quarter): 16 bytes), no cross-lane reduction; the waves'
// partial tiles are added through LDS and wave 0 writes the 64 pad columns of its 16 rows.
template <int P>
__global__ void __launch_bounds__(256) lora_down_kernel(bf16_t* __restrict__ A, int64_t lda, int K, int64_t rows, const bf16_t* __restrict__ D,
                                                        const float* __restrict__ dcolsum, const float* __restrict__ dbias, float eps, int ln,
                                                        const int* __restrict__ sets, int64_t rows_per_set) {
    __shared__ float red[3][2][4][64];                      // waves 1-3: C1 / C2 partials, [reg][lane]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    // blocks of 16 rows never cross a concept boundary: block = (batch row bb, 16-row piece of its rows_per_set rows)
    const int bps = (int)((rows_per_set + 15) >> 4);
    const int64_t bb = blockIdx.x / bps, m0 = bb * rows_per_set + (int64_t)(blockIdx.x - bb * bps) * 16;
    const int64_t mend = (bb + 1) * rows_per_set;           // (rows == batch rows x rows_per_set)
    const int set = sets[bb];
    const int64_t mrow = m0 + j < mend ? m0 + j : mend - 1;
    const bf16_t* xr = A + mrow * lda + kq * 8;
    const bf16_t* dr = D + ((int64_t)set * P + (j < P ? j : 0)) * K + kq * 8;
    lora_frag ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (__bf16)1.0f;
    lora_frag zero;
#pragma unroll
    for (int k = 0; k < 8; ++k) zero[k] = (__bf16)0.0f;
    f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = c1;
    const int nks = K >> 5;                                  // 32-wide k-steps; wave w takes every fourth
    for (int ks = w; ks < nks; ks += 4) {
        const lora_frag x = *(const lora_frag*)(xr + ks * 32);
        lora_frag d = j < P ? *(const lora_frag*)(dr + ks * 32) : (j == P ? ones : zero);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d, x, c1, 0, 0, 0);      // lane (column j = data row, rows 4 kq + r = down row)
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, c2, 0, 0, 0);
    }
    if (w) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[w - 1][0][r][lane] = c1[r]; red[w - 1][1][r][lane] = c2[r]; }
    }
    __syncthreads();
    if (w) return;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) { c1[r] += red[u][0][r][lane]; c2[r] += red[u][1][r][lane]; }
    // lane (j, kq) holds T[row j][4 kq + r]; s1 of row j sits in lane (j, P / 4) register P % 4, s2 (the Gram diagonal) in lane (j, j / 4) register j % 4
    float mean = 0.f, sd = 1.f;
    if (ln) {
        const float s1c = c1[P & 3];
        const float s2c = (j & 3) == 0 ? c2[0] : (j & 3) == 1 ? c2[1] : (j & 3) == 2 ? c2[2] : c2[3];
        const float s1 = __shfl(s1c, j + 16 * (P >> 2)), s2 = __shfl(s2c, j + 16 * (j >> 2));
        mean = s1 / (float)K;
        sd = sqrtf(fmaxf(s2 / (float)K - mean * mean, 0.f) + eps);            // 1 / rstd
    }
    if (m0 + j >= mend) return;
    // this lane's 4 values go to pad columns set * P + 4 kq .. (when 4 kq < P); every other group of 4 pad columns of the row gets zeros
    bf16_t* prow = A + (m0 + j) * lda + K;
    uint2 v = make_uint2(0u, 0u);
    if (4 * kq < P) {
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 4 * kq + r;
            t[r] = ln ? c1[r] - mean * dcolsum[set * P + q] + dbias[set * P + q] * sd : c1[r];
        }
        v = make_uint2(pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3]));
    }
    // 16 groups of 4 columns per row, 4 lanes (kq) per row: lane kq writes groups kq, kq + 4, kq + 8, kq + 12 -- its own values at group
    // (set * P) / 4 + kq' where kq' < P / 4 ... handled by value: group g holds values iff set * P / 4 <= g < (set * P + P) / 4
    const int g0 = (set * P) >> 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int g = kq + 4 * u;                            // this lane writes group g; the values for it live in lane kq' = g - g0 of this row
        const int src = g - g0;
        const bool has = src >= 0 && 4 * src < P;
        const unsigned vx = __shfl(v.x, j + 16 * (has ? src : 0)), vy = __shfl(v.y, j + 16 * (has ? src : 0));
        *(uint2*)(prow + 4 * g) = has ? make_uint2(vx, vy) : make_uint2(0u, 0u);
    }
}
}  // namespace

extern "C" int tmix_lora_down(void* A, int64_t lda, int K, int64_t rows, const void* D, int P, int nsets, const float* dcolsum,
                              const float* dbias, float eps, const int* sets, int64_t rows_per_set, void* stream) {
    if (!A || !D || !sets) TMIX_FAIL(TMIX_EINVAL, "lora_down: null pointer");
    if (rows <= 0 || K <= 0 || (K % 8) || lda < K + 64 || (lda % 8)) TMIX_FAIL(TMIX_ESHAPE, "lora_down: rows=%lld K=%d lda=%lld (rows carry 64 pad columns behind their K values)", (long long)rows, K, (long long)lda);
    if ((P != 4 && P != 12) || nsets < 1 || nsets * P > 64) TMIX_FAIL(TMIX_ESHAPE, "lora_down: P=%d (4 or 12) x nsets=%d must fit the 64 pad columns", P, nsets);
    if ((dcolsum == nullptr) != (dbias == nullptr)) TMIX_FAIL(TMIX_EINVAL, "lora_down: the folded-LayerNorm form needs dcolsum and dbias");
    if (!aligned16(A) || !aligned16(D)) TMIX_FAIL(TMIX_EALIGN, "lora_down: A / D must be 16-byte aligned");
    const int ln = dcolsum != nullptr;
    if (rows_per_set <= 0 || (rows % rows_per_set) || (K % 32)) TMIX_FAIL(TMIX_ESHAPE, "lora_down: rows=%lld must be a multiple of rows_per_set=%lld and K=%d of 32", (long long)rows, (long long)rows_per_set, K);
    const unsigned grid = (unsigned)((rows / rows_per_set) * ((rows_per_set + 15) / 16));
    if (P == 12) lora_down_kernel<12><<<grid, 256, 0, (hipStream_t)stream>>>((bf16_t*)A, lda, K, rows, (const bf16_t*)D, dcolsum, dbias, eps, ln, sets, rows_per_set);
    else lora_down_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>((bf16_t*)A, lda, K, rows, (const bf16_t*)D, dcolsum, dbias, eps, ln, sets, rows_per_set);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

// ------------------------------------------------------------------------------ temporal attention (frame axis, S <= 16)
// I2VGen-XL's TransformerTemporalModel attends over the FRAMES of one pixel: sequences of 16 tokens, head size 64, one (clip, pixel, head) item per wave,
// on the matrix cores: S^T = K Q^T is two v_mfma_f32_16x16x32_bf16 (operands straight from global memory: a lane's fragment is 16 contiguous bytes of
// its frame's row), the softmax runs over the four scores a lane holds and its three partner lanes (xor 16 / 32), and the exponentials ARE the B operand of
// O^T = V^T P^T (four v_mfma_f32_16x16x16_bf16, one per 16 channels) -- only V has to turn: its rows go through LDS and come back as four 2-byte reads per
// MFMA.  The round-3 form did the 2 x 16 x 16 x 64 multiply-adds of an item on the VALU (~175 us for the 86,016 x 5 items of the first level against
// ~100 us of HBM time for its 440 MB).
namespace {
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 ta_frag;
constexpr int TA_VLD = 68;                       // LDS row of V: 64 channels + 4 pad (136 B: the four frames a lane group reads sit 8 banks apart)
__global__ void __launch_bounds__(256) temporal_attn_kernel(const bf16_t* __restrict__ QKV, int64_t ld, bf16_t* __restrict__ O, int64_t ldo,
                                                            int frames, int64_t hw, int heads, int64_t items, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) bf16_t sV[4][16][TA_VLD];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + w;
    const bool live = item < items;
    const int C = heads * 64;
    const int r = lane & 15, g = lane >> 4;
    const int64_t ph = live ? item : 0;
    const int h = (int)(ph % heads);
    const int64_t cp = ph / heads;                                   // clip * hw + pixel
    const int64_t clip = cp / hw, pix = cp - clip * hw;
    const int64_t row = (clip * frames + (r < frames ? r : 0)) * hw + pix;      // token row of frame r (padding frames re-read frame 0: finite values)
    const bf16_t* q = QKV + row * ld + h * 64;
    // fragments of the score MFMAs: lane (r, g) holds channels [32 kb + 8 g, +8) of frame r -- K as A (rows = key frames), Q as B (columns = query frames)
    const ta_frag k0 = *(const ta_frag*)(q + C + g * 8), k1 = *(const ta_frag*)(q + C + 32 + g * 8);
    const ta_frag q0 = *(const ta_frag*)(q + g * 8), q1 = *(const ta_frag*)(q + 32 + g * 8);
    // V rows -> LDS (lane: frame r, channels [16 g, +16))
    {
        const uint4 va = *(const uint4*)(q + 2 * C + g * 16), vb = *(const uint4*)(q + 2 * C + g * 16 + 8);
        uint2* dst = (uint2*)&sV[w][r][g * 16];
        dst[0] = make_uint2(va.x, va.y); dst[1] = make_uint2(va.z, va.w); dst[2] = make_uint2(vb.x, vb.y); dst[3] = make_uint2(vb.z, vb.w);
    }
    f32x4 st = {0.f, 0.f, 0.f, 0.f};
    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, q0, st, 0, 0, 0);
    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, q1, st, 0, 0, 0);          // st[j] = K[4 g + j] . Q[r]
    float sc[4], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = (4 * g + j) < frames ? st[j] * scale_log2e : -INFINITY; mx = fmaxf(mx, sc[j]); }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = exp2f(sc[j] - mx); sum += sc[j]; }          // 0 for padded frames
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    short4_t pb;                                                                     // P^T as the B operand: column = query r, k = keys 4 g .. 4 g + 3
    {
        const unsigned lo = pack_bf2(sc[0], sc[1]), hi = pack_bf2(sc[2], sc[3]);
        pb = (short4_t){(short)(lo & 0xffff), (short)(lo >> 16), (short)(hi & 0xffff), (short)(hi >> 16)};
    }
    __syncthreads();
    bf16_t* dst = O + row * ldo + h * 64 + 4 * g;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        // V^T as the A operand: row = channel 16 db + r, k = keys 4 g .. 4 g + 3
        short4_t va;
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = (short)sV[w][4 * g + j][16 * db + r];
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, pb, o, 0, 0, 0);          // o[j] = O[query r][channel 16 db + 4 g + j] * sum
        if (live && r < frames) *(uint2*)(dst + 16 * db) = make_uint2(pack_bf2(o[0] * inv, o[1] * inv), pack_bf2(o[2] * inv, o[3] * inv));
    }
}
}  // namespace

extern "C" int tmix_temporal_attn(const void* QKV, int64_t ld, void* O, int64_t ldo, int clips, int frames, int64_t hw, int heads,
                                  float scale, void* stream) {
    if (!QKV || !O) TMIX_FAIL(TMIX_EINVAL, "temporal_attn: null pointer");
    if (clips < 1 || frames < 1 || frames > 16 || hw < 1 || heads < 1) TMIX_FAIL(TMIX_ESHAPE, "temporal_attn: clips=%d frames=%d (1..16) hw=%lld heads=%d", clips, frames, (long long)hw, heads);
    if (ld < 3 * heads * 64 || ldo < heads * 64 || (ld % 8) || (ldo % 8)) TMIX_FAIL(TMIX_ESHAPE, "temporal_attn: ld=%lld ldo=%lld for %d heads of 64", (long long)ld, (long long)ldo, heads);
    if (!aligned16(QKV) || !aligned16(O)) TMIX_FAIL(TMIX_EALIGN, "temporal_attn: pointers must be 16-byte aligned");
    const int64_t items = (int64_t)clips * hw * heads;
    const int64_t blocks = (items + 3) / 4;
    if (blocks > 0x7fffffff) TMIX_FAIL(TMIX_ESHAPE, "temporal_attn: grid too large");
    temporal_attn_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)QKV, ld, (bf16_t*)O, ldo, frames, hw, heads, items,
                                                                            scale * 1.4426950408889634f);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_softmax_rows(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int cols, float scale, void* stream) {
    if (!S || !P) TMIX_FAIL(TMIX_EINVAL, "softmax_rows: null pointer");
    if (rows <= 0 || cols <= 0 || (cols % 4) || (ld_s % 4) || (ld_p % 4)) TMIX_FAIL(TMIX_ESHAPE, "softmax_rows: rows=%lld cols=%d (cols, ld %% 4 == 0)", (long long)rows, cols);
    if (!aligned16(S) || (((uintptr_t)P) & 7)) TMIX_FAIL(TMIX_EALIGN, "softmax_rows: pointer alignment");
    softmax_rows_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(S, ld_s, (bf16_t*)P, ld_p, cols, scale * 1.4426950408889634f, 0, cols);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_softmax_rows_causal(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int cols, float scale,
                                        int seq, void* stream) {
    if (!S || !P) TMIX_FAIL(TMIX_EINVAL, "softmax_rows_causal: null pointer");
    if (rows <= 0 || cols <= 0 || seq <= 0 || seq > cols || (rows % seq) || (cols % 4) || (ld_s % 4) || (ld_p % 4))
        TMIX_FAIL(TMIX_ESHAPE, "softmax_rows_causal: rows=%lld cols=%d seq=%d (rows %% seq == 0, seq <= cols, cols/ld %% 4 == 0)", (long long)rows, cols, seq);
    if (!aligned16(S) || (((uintptr_t)P) & 7)) TMIX_FAIL(TMIX_EALIGN, "softmax_rows_causal: pointer alignment");
    softmax_rows_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(S, ld_s, (bf16_t*)P, ld_p, cols, scale * 1.4426950408889634f, seq, cols);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_softmax_rows_masked(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int cols, int valid, float scale,
                                        void* stream) {
    if (!S || !P) TMIX_FAIL(TMIX_EINVAL, "softmax_rows_masked: null pointer");
    if (rows <= 0 || cols <= 0 || valid < 1 || valid > cols || (cols % 4) || (ld_s % 4) || (ld_p % 4))
        TMIX_FAIL(TMIX_ESHAPE, "softmax_rows_masked: rows=%lld cols=%d valid=%d", (long long)rows, cols, valid);
    if (!aligned16(S) || (((uintptr_t)P) & 7)) TMIX_FAIL(TMIX_EALIGN, "softmax_rows_masked: pointer alignment");
    softmax_rows_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(S, ld_s, (bf16_t*)P, ld_p, cols, scale * 1.4426950408889634f, 0, valid);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_affine_clamp(const float* x, float* y, int64_t n, float scale, float shift, float lo, float hi, void* stream) {
    if (!x || !y) TMIX_FAIL(TMIX_EINVAL, "affine_clamp: null pointer");
    if (n <= 0) TMIX_FAIL(TMIX_ESHAPE, "affine_clamp: empty");
    int64_t nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
    affine_clamp_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(x, y, n, scale, shift, lo, hi);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_timestep_embedding(const float* values, float* out, int count, int dim, void* stream) {
    if (!values || !out) TMIX_FAIL(TMIX_EINVAL, "timestep_embedding: null pointer");
    if (count <= 0 || dim <= 0 || (dim & 1)) TMIX_FAIL(TMIX_ESHAPE, "timestep_embedding: count=%d dim=%d", count, dim);
    const int n = count * (dim / 2);
    timestep_embedding_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(values, out, count, dim);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

// rows beyond 16 go out in further launches of 16 (co-batched seeds: B = 32 rows of time / text embeddings)
static int linear_small_launch(const float* in, const void* W, const float* bias, const float* add, float* out, int M, int N, int K,
                               int act_in, int act_out, const int* secs, int nsec, hipStream_t st) {
    for (int m0 = 0; m0 < M; m0 += 16) {
        const int m = M - m0 < 16 ? M - m0 : 16;
        const float* in_c = in + (int64_t)m0 * K;
        const float* add_c = add ? add + (int64_t)m0 * N : nullptr;
        // sections: every section is a dense [M][width] matrix, so a row chunk starts m0 * width into each -- the kernel adds
        // secs[s] * M itself; the plain form is one [M][N] matrix
        float* out_c = secs ? out : out + (int64_t)m0 * N;
        if (m <= 4) linear_small_kernel<4><<<(N + 3) / 4, 256, 0, st>>>(in_c, (const bf16_t*)W, bias, add_c, out_c, m, N, K, act_in, act_out, secs, nsec, M, m0);
        else        linear_small_kernel<16><<<(N + 3) / 4, 256, 0, st>>>(in_c, (const bf16_t*)W, bias, add_c, out_c, m, N, K, act_in, act_out, secs, nsec, M, m0);
        TMIX_LAUNCH_CHECK();
    }
    return TMIX_OK;
}

extern "C" int tmix_linear_small(const float* in, const void* W, const float* bias, const float* add, float* out,
                                 int M, int N, int K, int act_in, int act_out, void* stream) {
    if (!in || !W || !out) TMIX_FAIL(TMIX_EINVAL, "linear_small: null pointer");
    if (M <= 0 || M > 256 || N <= 0 || K <= 0 || (K % 8)) TMIX_FAIL(TMIX_ESHAPE, "linear_small: M=%d (1..256) N=%d K=%d (K %% 8 == 0)", M, N, K);
    if (!aligned16(in) || !aligned16(W)) TMIX_FAIL(TMIX_EALIGN, "linear_small: in/W must be 16-byte aligned");
    return linear_small_launch(in, W, bias, add, out, M, N, K, act_in, act_out, nullptr, 0, (hipStream_t)stream);
}

extern "C" int tmix_linear_small_sections(const float* in, const void* W, const float* bias, float* out, int M, int N, int K,
                                          int act_in, const int* sec_starts, int nsec, void* stream) {
    if (!in || !W || !out || !sec_starts) TMIX_FAIL(TMIX_EINVAL, "linear_small_sections: null pointer");
    if (M <= 0 || M > 256 || N <= 0 || K <= 0 || (K % 8) || nsec < 1) TMIX_FAIL(TMIX_ESHAPE, "linear_small_sections: M=%d (1..256) N=%d K=%d (K %% 8 == 0) nsec=%d", M, N, K, nsec);
    if (!aligned16(in) || !aligned16(W)) TMIX_FAIL(TMIX_EALIGN, "linear_small_sections: in/W must be 16-byte aligned");
    return linear_small_launch(in, W, bias, nullptr, out, M, N, K, act_in, 0, sec_starts, nsec, (hipStream_t)stream);
}

extern "C" int tmix_conv_in(const float* x_nchw, const float* w_ohwi, const float* bias, void* y_nhwc,
                            int B, int Cin, int H, int W, int Cout, void* stream) {
    return tmix_conv_in_pre(x_nchw, w_ohwi, bias, y_nhwc, B, Cin, H, W, Cout, nullptr, nullptr, stream);
}

extern "C" int tmix_conv_in_pre(const float* x_nchw, const float* w_ohwi, const float* bias, void* y_nhwc,
                                int B, int Cin, int H, int W, int Cout, const float* pre_w, const float* pre_b, void* stream) {
    if (!x_nchw || !w_ohwi || !y_nhwc) TMIX_FAIL(TMIX_EINVAL, "conv_in: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) TMIX_FAIL(TMIX_ESHAPE, "conv_in: bad shape");
    if (!aligned16(y_nhwc)) TMIX_FAIL(TMIX_EALIGN, "conv_in: output must be 16-byte aligned");
    if (Cout % 32) TMIX_FAIL(TMIX_ESHAPE, "conv_in: Cout=%d must be a multiple of 32", Cout);
    if (Cin != 3 && Cin != 4 && Cin != 8) TMIX_FAIL(TMIX_ESHAPE, "conv_in: Cin=%d (3: RGB image, 4: image latent, 8: video latent + image-latent features)", Cin);
    if (Cin != 4 && pre_w) TMIX_FAIL(TMIX_EINVAL, "conv_in: the latent pre-map is defined for 4 channels");
    if ((Cout * 9 * Cin) % 4) TMIX_FAIL(TMIX_ESHAPE, "conv_in: Cout*9*Cin must be a multiple of 4");
    if (!aligned16(w_ohwi)) TMIX_FAIL(TMIX_EALIGN, "conv_in: weights must be 16-byte aligned");
    const int64_t npix = (int64_t)B * H * W;
    const int64_t ntiles = (npix + 63) / 64;
    const int smem = Cout * 9 * Cin * 4;
    const int per_cu = smem > 80 * 1024 ? 1 : (smem > 52 * 1024 ? 2 : 3);            // resident workgroups per CU (LDS)
    const unsigned nb = (unsigned)(ntiles < 256 * per_cu ? ntiles : 256 * per_cu);
    if (smem > 150 * 1024) TMIX_FAIL(TMIX_ESHAPE, "conv_in: Cout=%d too large for the LDS weight stage", Cout);
    static int attr_smem[3] = {0, 0, 0};
    const int slot = Cin == 8 ? 2 : (Cin == 4 ? 1 : 0);
    if (smem > 64 * 1024 && smem > attr_smem[slot]) {
        hipError_t e = Cin == 8 ? hipFuncSetAttribute((const void*)conv_in_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)
                     : Cin == 4 ? hipFuncSetAttribute((const void*)conv_in_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)
                                : hipFuncSetAttribute((const void*)conv_in_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_smem[slot] = smem;
    }
    PreMap pm = {};
    if (pre_w) {
        pm.on = 1;
        for (int i = 0; i < 16; ++i) pm.w[i] = pre_w[i];
        for (int i = 0; i < 4; ++i) pm.b[i] = pre_b ? pre_b[i] : 0.f;
    }
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 8) conv_in_kernel<8><<<nb, 256, smem, st>>>(x_nchw, w_ohwi, bias, (bf16_t*)y_nhwc, B, H, W, Cout, pm);
    else if (Cin == 3) conv_in_kernel<3><<<nb, 256, smem, st>>>(x_nchw, w_ohwi, bias, (bf16_t*)y_nhwc, B, H, W, Cout, pm);
    else          conv_in_kernel<4><<<nb, 256, smem, st>>>(x_nchw, w_ohwi, bias, (bf16_t*)y_nhwc, B, H, W, Cout, pm);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

extern "C" int tmix_conv_out(const void* x_nhwc, const void* w_ohwi, const float* bias, float* y_nchw,
                             int B, int Cin, int H, int W, int Cout, void* stream) {
    if (!x_nhwc || !w_ohwi || !y_nchw) TMIX_FAIL(TMIX_EINVAL, "conv_out: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 16 || (Cin % 32)) TMIX_FAIL(TMIX_ESHAPE, "conv_out: Cin=%d (%%32) Cout=%d (<=16)", Cin, Cout);
    if (!aligned16(x_nhwc) || !aligned16(w_ohwi)) TMIX_FAIL(TMIX_EALIGN, "conv_out: pointers must be 16-byte aligned");
    const int64_t npix = (int64_t)B * H * W;
    const unsigned nb = (unsigned)((npix + 63) / 64);
    conv_out_kernel<<<nb, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x_nhwc, (const bf16_t*)w_ohwi, bias, y_nchw, B, H, W, Cin, Cout);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}
