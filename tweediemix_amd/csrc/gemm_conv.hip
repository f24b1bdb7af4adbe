// gemm_conv.hip -- bf16 MFMA GEMM (C = A * W^T) and 3x3 NHWC implicit-GEMM convolution for gfx950.
//
// One templated mainloop serves both: a 128x128 output tile per 256-thread workgroup (4 waves, 2x2,
// 64x64 per wave = 4x4 fragments of v_mfma_f32_16x16x32_bf16), BK = 64, operands staged HBM->LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a double-buffered, XOR-swizzled LDS image:
// the LDS destination of an LDS-DMA is lane-linear, so the swizzle is applied to each lane's SOURCE
// address (16-byte chunk c of row r is stored at chunk position c ^ (r & 7)) and undone on the
// ds_read_b128 side -- conflict-free for the 16-lane read groups of ds_read_b128.
// The convolution differs only in how a lane finds its source address (im2col on the fly: K-tile kt
// is tap kt / (Cin/64), channels (kt % (Cin/64))*64..+63 of the shifted pixel; padding taps read a
// zero page).  Fused epilogues: bias, per-row-group bias (time embedding), residual add, GEGLU,
// and a transposed store (V^T for the attention kernel).
//
// MFMA roofline: 2*M*N*K flops per launch against the 2.5 PFLOP/s dense bf16 peak.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;      // double buffer = 64 KiB -> 2 workgroups per CU

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;

__device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];   // 256 B of zeros (conv padding taps)

struct Params {
    const bf16_t* A; int64_t lda, strideA;
    const bf16_t* W; int64_t ldw, strideW;
    bf16_t* C; int64_t ldc, strideC;
    const float* bias; int64_t strideBias;
    const bf16_t* R; int64_t ldr, strideR;
    const float* rgb; int rows_per_group;
    bf16_t* Ct; int64_t ldct, strideCt; int n_trans_begin;
    int M, N, K, tiles_m, tiles_n, epilogue;
    // convolution geometry (CONV only)
    int H, Wd, Cin, Ho, Wo, mode;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <bool SWAP>
__device__ __forceinline__ void mma_tile(const char* sA, const char* sW, int wr, int wc, int fr, int fg,
                                         f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fg) ^ (fr & 7)) << 4;
        frag_ab a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *(const frag_ab*)(sA + (wr * 64 + i * 16 + fr) * 128 + sw);
            b[i] = *(const frag_ab*)(sW + (wc * 64 + i * 16 + fr) * 128 + sw);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if constexpr (SWAP)   // D[n][m]: lane holds 4 consecutive n for one m (row-major C stores)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
                else                  // D[m][n]: lane holds 4 consecutive m for one n (transposed stores)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
    }
}

template <int CONV>
__global__ void __launch_bounds__(256, 2) gemm_conv_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int fr = lane & 15, fg = lane >> 4;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = blockIdx.y;

    const bf16_t* Ab = p.A + (int64_t)bz * p.strideA;
    const bf16_t* Wb = p.W + (int64_t)bz * p.strideW;

    // ---- per-lane staging sources: 4 rounds x (8 rows x 8 chunks) per wave-instruction
    const int lrow = lane >> 3;                       // row within the 8-row group
    const int schunk = ((lane & 7) ^ lrow) * 8;       // swizzled source chunk (elements)
    const bf16_t* wsrc[4];
    const bf16_t* asrc[4];
    int pb[4], py[4], px[4];                          // conv: decoded output pixel per round
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = (r * 4 + w) * 8 + lrow;
        int n = n0 + row; if (n > p.N - 1) n = p.N - 1;
        wsrc[r] = Wb + (int64_t)n * p.ldw + schunk;
        int m = m0 + row; if (m > p.M - 1) m = p.M - 1;
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo;
            pb[r] = m / hw; const int rem = m - pb[r] * hw;
            py[r] = rem / p.Wo; px[r] = rem - py[r] * p.Wo;
            asrc[r] = nullptr;
        } else {
            asrc[r] = Ab + (int64_t)m * p.lda + schunk;
        }
    }

    const int nk = p.K / BK;
    int tap = 0, cc = 0;                              // conv K-tile cursor: tap (0..8), 64-channel chunk
    const int cpt = CONV ? p.Cin / BK : 1;

    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * STAGE_BYTES;
        char* sW = sA + TILE_BYTES;
        int ky = 0, kx = 0;
        if constexpr (CONV) { ky = tap / 3; kx = tap - ky * 3; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int off = (r * 4 + w) * 1024;
            const bf16_t* ga;
            if constexpr (CONV) {
                int iy, ix; bool ok;
                if (p.mode == TMIX_CONV_S1)      { iy = py[r] + ky - 1;     ix = px[r] + kx - 1;     ok = (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd); }
                else if (p.mode == TMIX_CONV_S2) { iy = 2 * py[r] + ky - 1; ix = 2 * px[r] + kx - 1; ok = (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd); }
                else { const int uy = py[r] + ky - 1, ux = px[r] + kx - 1;   // conv over the nearest-x2 upsampled image
                       ok = (uy >= 0) & (uy < 2 * p.H) & (ux >= 0) & (ux < 2 * p.Wd); iy = uy >> 1; ix = ux >> 1; }
                ga = ok ? Ab + ((int64_t)(pb[r] * p.H + iy) * p.Wd + ix) * p.Cin + cc * BK + schunk
                        : (const bf16_t*)g_zero_page + schunk;
            } else {
                ga = asrc[r] + (int64_t)kt * BK;
            }
            glds16(ga, sA + off);
            glds16(wsrc[r] + (int64_t)kt * BK, sW + off);
        }
        if constexpr (CONV) { if (++cc == cpt) { cc = 0; ++tap; } }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool trans = (p.n_trans_begin >= 0) && (n0 >= p.n_trans_begin);

    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sA = smem + cur * STAGE_BYTES;
        const char* sW = sA + TILE_BYTES;
        if (trans) mma_tile<false>(sA, sW, wr, wc, fr, fg, acc);
        else       mma_tile<true>(sA, sW, wr, wc, fr, fg, acc);
        __syncthreads();
        cur ^= 1;
    }

    // ---------------------------------------------------------------- epilogue
    const float* bias = p.bias ? p.bias + (int64_t)bz * p.strideBias : nullptr;
    if (trans) {
        bf16_t* Ct = p.Ct + (int64_t)bz * p.strideCt;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int m = m0 + wr * 64 + mi * 16 + fg * 4;
                const int n = n0 + wc * 64 + ni * 16 + fr;
                if (n >= p.N || m >= p.M) continue;
                const float bv = bias ? bias[n] : 0.f;
                bf16_t* dst = Ct + (int64_t)(n - p.n_trans_begin) * p.ldct + m;
                if (m + 3 < p.M && ((p.ldct & 3) == 0)) {
                    uint2 v;
                    v.x = pack_bf2(acc[mi][ni][0] + bv, acc[mi][ni][1] + bv);
                    v.y = pack_bf2(acc[mi][ni][2] + bv, acc[mi][ni][3] + bv);
                    *(uint2*)dst = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (m + r < p.M) dst[r] = f2bf(acc[mi][ni][r] + bv);
                }
            }
        return;
    }

    bf16_t* Cb = p.C + (int64_t)bz * p.strideC;
    const bf16_t* Rb = p.R ? p.R + (int64_t)bz * p.strideR : nullptr;
    if (p.epilogue == TMIX_EPI_GEGLU) {
        // weight rows are interleaved in 16-row groups [value_j | gate_j]: even fragments hold the
        // value half, odd fragments the gate half of the same 16 output columns.
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wr * 64 + mi * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const int nv = n0 + wc * 64 + nj * 32 + fg * 4;        // value columns (weight-row index)
                if (nv >= p.N) continue;
                const int no = (n0 + wc * 64) / 2 + nj * 16 + fg * 4;  // output column
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = acc[mi][2 * nj][r], g = acc[mi][2 * nj + 1][r];
                    if (bias) { a += bias[nv + r]; g += bias[nv + 16 + r]; }
                    o[r] = a * gelu_erf_f(g);
                }
                uint2 v; v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]);
                *(uint2*)(Cb + (int64_t)m * p.ldc + no) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr * 64 + mi * 16 + fr;
        if (m >= p.M) continue;
        const float* rg = p.rgb ? p.rgb + (int64_t)(m / p.rows_per_group) * p.N : nullptr;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wc * 64 + ni * 16 + fg * 4;
            if (n >= p.N) continue;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[mi][ni][r];
            if (bias) { const float4 b4 = *(const float4*)(bias + n); o[0] += b4.x; o[1] += b4.y; o[2] += b4.z; o[3] += b4.w; }
            if (rg)   { const float4 b4 = *(const float4*)(rg + n);   o[0] += b4.x; o[1] += b4.y; o[2] += b4.z; o[3] += b4.w; }
            if (Rb) {
                const uint2 rv = *(const uint2*)(Rb + (int64_t)m * p.ldr + n);
                o[0] += bf2f((bf16_t)(rv.x & 0xffff)); o[1] += bf2f((bf16_t)(rv.x >> 16));
                o[2] += bf2f((bf16_t)(rv.y & 0xffff)); o[3] += bf2f((bf16_t)(rv.y >> 16));
            }
            uint2 v; v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]);
            *(uint2*)(Cb + (int64_t)m * p.ldc + n) = v;
        }
    }
}

template <int CONV>
int launch(const Params& p, int batch, hipStream_t st) {
    static bool attr_set = false;   // idempotent; racing threads set the same value
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_conv_kernel<CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n, batch, 1);
    gemm_conv_kernel<CONV><<<grid, 256, SMEM_BYTES, st>>>(p);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

}  // namespace

extern "C" int tmix_gemm_bf16(const tmix_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->W) TMIX_FAIL(TMIX_EINVAL, "gemm: null descriptor/operand");
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) TMIX_FAIL(TMIX_ESHAPE, "gemm: empty problem M=%d N=%d K=%d batch=%d", d->M, d->N, d->K, d->batch);
    if (d->K % BK) TMIX_FAIL(TMIX_ESHAPE, "gemm: K=%d must be a multiple of %d", d->K, BK);
    if (d->N % 4) TMIX_FAIL(TMIX_ESHAPE, "gemm: N=%d must be a multiple of 4", d->N);
    if ((d->lda % 8) || (d->ldw % 8) || (d->strideA % 8) || (d->strideW % 8)) TMIX_FAIL(TMIX_EALIGN, "gemm: lda/ldw/strides must be multiples of 8 elements");
    if (!aligned16(d->A) || !aligned16(d->W)) TMIX_FAIL(TMIX_EALIGN, "gemm: A/W must be 16-byte aligned");
    const bool has_trans = d->n_trans_begin >= 0 && d->n_trans_begin < d->N;
    if (has_trans && (!d->Ct || (d->n_trans_begin % BN))) TMIX_FAIL(TMIX_EINVAL, "gemm: transposed region needs Ct and n_trans_begin %% %d == 0", BN);
    if ((!has_trans || d->n_trans_begin > 0) && !d->C) TMIX_FAIL(TMIX_EINVAL, "gemm: null C");
    if (d->C && ((d->ldc % 4) || (((uintptr_t)d->C) & 7) || (d->strideC % 4))) TMIX_FAIL(TMIX_EALIGN, "gemm: C must be 8-byte aligned with ldc %% 4 == 0");
    if (d->residual && ((d->ldr % 4) || (((uintptr_t)d->residual) & 7))) TMIX_FAIL(TMIX_EALIGN, "gemm: residual alignment");
    if (d->bias && (((uintptr_t)d->bias) & 15)) TMIX_FAIL(TMIX_EALIGN, "gemm: bias must be 16-byte aligned");
    if (d->rowgroup_bias && (d->rows_per_group <= 0 || (((uintptr_t)d->rowgroup_bias) & 15))) TMIX_FAIL(TMIX_EINVAL, "gemm: rowgroup_bias needs rows_per_group > 0 and 16-byte alignment");
    if (d->epilogue == TMIX_EPI_GEGLU && ((d->N % 32) || has_trans || d->residual || d->rowgroup_bias)) TMIX_FAIL(TMIX_EINVAL, "gemm: GEGLU needs N %% 32 == 0 and no residual/transposed region");
    if (d->epilogue != TMIX_EPI_NONE && d->epilogue != TMIX_EPI_GEGLU) TMIX_FAIL(TMIX_EINVAL, "gemm: bad epilogue %d", d->epilogue);
    Params p = {};
    p.A = (const bf16_t*)d->A; p.lda = d->lda; p.strideA = d->strideA;
    p.W = (const bf16_t*)d->W; p.ldw = d->ldw; p.strideW = d->strideW;
    p.C = (bf16_t*)d->C; p.ldc = d->ldc; p.strideC = d->strideC;
    p.bias = d->bias; p.strideBias = d->strideBias;
    p.R = (const bf16_t*)d->residual; p.ldr = d->ldr; p.strideR = d->strideR;
    p.rgb = d->rowgroup_bias; p.rows_per_group = d->rows_per_group;
    p.Ct = (bf16_t*)d->Ct; p.ldct = d->ldct; p.strideCt = d->strideCt;
    p.n_trans_begin = has_trans ? d->n_trans_begin : -1;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.tiles_m = (d->M + BM - 1) / BM; p.tiles_n = (d->N + BN - 1) / BN;
    p.epilogue = d->epilogue;
    return launch<0>(p, d->batch, (hipStream_t)stream);
}

extern "C" int tmix_conv3x3_nhwc(const tmix_conv_desc* d, void* stream) {
    if (!d || !d->X || !d->Wt || !d->Y) TMIX_FAIL(TMIX_EINVAL, "conv3x3: null descriptor/operand");
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: empty problem");
    if (d->Cin % BK) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: Cin=%d must be a multiple of %d", d->Cin, BK);
    if (d->Cout % 4) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: Cout=%d must be a multiple of 4", d->Cout);
    if (d->mode < TMIX_CONV_S1 || d->mode > TMIX_CONV_UP2) TMIX_FAIL(TMIX_EINVAL, "conv3x3: bad mode %d", d->mode);
    if (d->mode == TMIX_CONV_S2 && ((d->H | d->W) & 1)) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: stride-2 needs even H,W");
    if (!aligned16(d->X) || !aligned16(d->Wt) || (((uintptr_t)d->Y) & 7)) TMIX_FAIL(TMIX_EALIGN, "conv3x3: pointer alignment");
    if ((d->bias && (((uintptr_t)d->bias) & 15)) || (d->batch_bias && (((uintptr_t)d->batch_bias) & 15))) TMIX_FAIL(TMIX_EALIGN, "conv3x3: bias alignment");
    Params p = {};
    p.H = d->H; p.Wd = d->W; p.Cin = d->Cin; p.mode = d->mode;
    p.Ho = d->mode == TMIX_CONV_S2 ? d->H / 2 : (d->mode == TMIX_CONV_UP2 ? d->H * 2 : d->H);
    p.Wo = d->mode == TMIX_CONV_S2 ? d->W / 2 : (d->mode == TMIX_CONV_UP2 ? d->W * 2 : d->W);
    const int64_t M = (int64_t)d->B * p.Ho * p.Wo;
    if (M > 0x7fffffff / 4) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: too many output pixels");
    p.A = (const bf16_t*)d->X;
    p.W = (const bf16_t*)d->Wt; p.ldw = 9 * (int64_t)d->Cin;
    p.C = (bf16_t*)d->Y; p.ldc = d->Cout;
    p.bias = d->bias;
    p.R = (const bf16_t*)d->residual; p.ldr = d->Cout;
    p.rgb = d->batch_bias; p.rows_per_group = p.Ho * p.Wo;
    p.n_trans_begin = -1;
    p.M = (int)M; p.N = d->Cout; p.K = 9 * d->Cin;
    p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
    p.epilogue = TMIX_EPI_NONE;
    return launch<1>(p, 1, (hipStream_t)stream);
}
