// tweedie_step.hip -- fused CFG + Tweedie x0 + mask blend + DDIM update (one HBM pass).
// Follows fusion_generation/fusion_sampling.py:376-385,392-403,406-412,424-430,471-472.
// HBM-bound: per element it reads (rows) eps values + x + K mask values and writes 1-2 floats.
// Compiled with -ffp-contract=off so fp32 results are bit-identical to the numpy oracle.
#include "common.h"

namespace {

template <int DT> struct EpsT;
template <> struct EpsT<TMIX_F32>  { typedef float  T; static __device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; } };
template <> struct EpsT<TMIX_F16>  { typedef __half T; static __device__ __forceinline__ float ld(const __half* p, int64_t i) { return __half2float(p[i]); } };
template <> struct EpsT<TMIX_BF16> { typedef bf16_t T; static __device__ __forceinline__ float ld(const bf16_t* p, int64_t i) { return bf2f(p[i]); } };

// rounding point of the reference's autocast path (only when eps is fp16)
template <int DT> __device__ __forceinline__ float rnd(float v) {
    if constexpr (DT == TMIX_F16) {
        // torch evaluates fp16 elementwise ops in fp32 and rounds the fp32 RESULT to fp16 (two roundings).
        // The empty asm makes the fp32 value opaque so LLVM cannot fold mul+convert into the single-rounding
        // v_fma_mixlo_f16, which differs from the reference on near-ties.
        asm volatile("" : "+v"(v));
        return __half2float(__float2half_rn(v));
    } else return v;
}

template <int DT> __device__ __forceinline__ float cfg(float eu, float ec, float g) {
    const float d = rnd<DT>(ec - eu);
    const float gd = rnd<DT>(g * d);
    return rnd<DT>(eu + gd);
}

template <int DT, int MODE>
__global__ void __launch_bounds__(256)
tweedie_step_kernel(const float* __restrict__ x, const typename EpsT<DT>::T* __restrict__ eps,
                    const float* __restrict__ masks, float* __restrict__ out_x, float* __restrict__ out_x0,
                    int K, int64_t n, int64_t hw, float g, float sa, float s1, float sa_n, float s1_n, int is_last) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float xv = x[i];
        const float eu = EpsT<DT>::ld(eps, i);
        float x0;
        if constexpr (MODE == TMIX_STEP_FUSION) {
            const int64_t p = i % hw;
            x0 = 0.0f;
            for (int c = 0; c < K; ++c) {
                const float e = cfg<DT>(eu, EpsT<DT>::ld(eps, (int64_t)(1 + c) * n + i), g);
                const float t = (xv - rnd<DT>(s1 * e)) / sa;
                x0 = x0 + masks[(int64_t)c * hw + p] * t;
            }
        } else if constexpr (MODE == TMIX_STEP_PLAIN) {
            const float e = cfg<DT>(eu, EpsT<DT>::ld(eps, n + i), g);
            x0 = (xv - rnd<DT>(s1 * e)) / sa;
        } else {   // RESAMPLE: (K-1)*x0_multi - sum_{c<K-1} x0_single_c
            const float em = cfg<DT>(eu, EpsT<DT>::ld(eps, n + i), g);
            x0 = (float)(K - 1) * ((xv - rnd<DT>(s1 * em)) / sa);
            for (int c = 0; c < K - 1; ++c) {
                const float e = cfg<DT>(eu, EpsT<DT>::ld(eps, (int64_t)(2 + c) * n + i), g);
                x0 = x0 - (xv - rnd<DT>(s1 * e)) / sa;
            }
        }
        const float moved = sa_n * x0 + rnd<DT>(s1_n * eu);
        out_x[i] = is_last ? x0 : moved;
        if (out_x0) out_x0[i] = x0;
    }
}

template <int DT>
int launch(const float* x, const void* eps, const float* masks, float* out_x, float* out_x0, int K, int64_t n,
           int64_t hw, int mode, float g, float sa, float s1, float sa_n, float s1_n, int is_last, hipStream_t st) {
    typedef typename EpsT<DT>::T T;
    const int threads = 256;
    int64_t blocks = (n + threads - 1) / threads;
    if (blocks > 2048) blocks = 2048;
    const T* e = (const T*)eps;
    switch (mode) {
    case TMIX_STEP_FUSION:
        tweedie_step_kernel<DT, TMIX_STEP_FUSION><<<blocks, threads, 0, st>>>(x, e, masks, out_x, out_x0, K, n, hw, g, sa, s1, sa_n, s1_n, is_last); break;
    case TMIX_STEP_PLAIN:
        tweedie_step_kernel<DT, TMIX_STEP_PLAIN><<<blocks, threads, 0, st>>>(x, e, masks, out_x, out_x0, K, n, hw, g, sa, s1, sa_n, s1_n, is_last); break;
    default:
        tweedie_step_kernel<DT, TMIX_STEP_RESAMPLE><<<blocks, threads, 0, st>>>(x, e, masks, out_x, out_x0, K, n, hw, g, sa, s1, sa_n, s1_n, is_last); break;
    }
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

}  // namespace

extern "C" int tmix_fused_tweedie_step(const float* x, const void* eps, int eps_dtype, const float* masks,
                                       float* out_x, float* out_x0, int K, int channels, int64_t hw, int mode,
                                       float g, float sa, float s1, float sa_next, float s1_next, int is_last,
                                       void* stream) {
    if (!x || !eps || !out_x) TMIX_FAIL(TMIX_EINVAL, "tweedie_step: null pointer");
    if (mode < TMIX_STEP_FUSION || mode > TMIX_STEP_RESAMPLE) TMIX_FAIL(TMIX_EINVAL, "tweedie_step: bad mode %d", mode);
    if (mode == TMIX_STEP_FUSION && (!masks || K < 1)) TMIX_FAIL(TMIX_EINVAL, "tweedie_step: FUSION needs masks and K>=1");
    if (mode == TMIX_STEP_RESAMPLE && K < 1) TMIX_FAIL(TMIX_EINVAL, "tweedie_step: RESAMPLE needs K>=1");
    if (channels < 1 || hw < 1) TMIX_FAIL(TMIX_ESHAPE, "tweedie_step: empty latent (channels=%d hw=%lld)", channels, (long long)hw);
    if (!(sa > 0.0f)) TMIX_FAIL(TMIX_EINVAL, "tweedie_step: sqrt(alpha) must be > 0");
    const int64_t n = (int64_t)channels * hw;
    hipStream_t st = (hipStream_t)stream;
    switch (eps_dtype) {
    case TMIX_F32:  return launch<TMIX_F32>(x, eps, masks, out_x, out_x0, K, n, hw, mode, g, sa, s1, sa_next, s1_next, is_last, st);
    case TMIX_F16:  return launch<TMIX_F16>(x, eps, masks, out_x, out_x0, K, n, hw, mode, g, sa, s1, sa_next, s1_next, is_last, st);
    case TMIX_BF16: return launch<TMIX_BF16>(x, eps, masks, out_x, out_x0, K, n, hw, mode, g, sa, s1, sa_next, s1_next, is_last, st);
    }
    TMIX_FAIL(TMIX_EINVAL, "tweedie_step: bad eps dtype %d", eps_dtype);
}
