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

// DEV = 1: the step coefficients come from a device buffer prm = {t, sa, s1, sa_next, s1_next, is_last, g} (so ONE captured
// launch serves every timestep of a hipGraph replay) and blockIdx.y walks co-batched seeds (x / out: [seeds][n], eps:
// [seeds][rows][n], masks: [seeds][K][hw] or shared).  x may alias out_x: element i is read before it is written.
template <int DT, int MODE, int DEV = 0>
__global__ void __launch_bounds__(256)
tweedie_step_kernel(const float* x, const typename EpsT<DT>::T* __restrict__ eps,
                    const float* __restrict__ masks, float* out_x, float* __restrict__ out_x0,
                    int K, int64_t n, int64_t hw, float g, float sa, float s1, float sa_n, float s1_n, int is_last,
                    const float* __restrict__ prm = nullptr, int rows = 0, int64_t mask_seed_stride = 0) {
    if constexpr (DEV) {
        sa = prm[1]; s1 = prm[2]; sa_n = prm[3]; s1_n = prm[4]; is_last = prm[5] != 0.0f; g = prm[6];
        const int64_t sd = blockIdx.y;
        x += sd * n; out_x += sd * n; eps += sd * rows * n; masks += sd * mask_seed_stride;
        if (out_x0) out_x0 += sd * n;
    }
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

template <int DT>
int launch_dev(const float* x, const void* eps, const float* masks, int64_t mss, float* out_x, float* out_x0, int K, int64_t n,
               int64_t hw, int mode, int rows, int seeds, const float* prm, hipStream_t st) {
    typedef typename EpsT<DT>::T T;
    int64_t bx = (n + 255) / 256; if (bx > 2048) bx = 2048;
    const dim3 grid((unsigned)bx, (unsigned)seeds);
    const T* e = (const T*)eps;
    switch (mode) {
    case TMIX_STEP_FUSION:
        tweedie_step_kernel<DT, TMIX_STEP_FUSION, 1><<<grid, 256, 0, st>>>(x, e, masks, out_x, out_x0, K, n, hw, 0.f, 1.f, 0.f, 1.f, 0.f, 0, prm, rows, mss); break;
    case TMIX_STEP_PLAIN:
        tweedie_step_kernel<DT, TMIX_STEP_PLAIN, 1><<<grid, 256, 0, st>>>(x, e, masks, out_x, out_x0, K, n, hw, 0.f, 1.f, 0.f, 1.f, 0.f, 0, prm, rows, mss); break;
    default:
        tweedie_step_kernel<DT, TMIX_STEP_RESAMPLE, 1><<<grid, 256, 0, st>>>(x, e, masks, out_x, out_x0, K, n, hw, 0.f, 1.f, 0.f, 1.f, 0.f, 0, prm, rows, mss); break;
    }
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

// head of a captured denoising step: every seed's latent is broadcast over its UNet batch rows and the timestep
// (prm[0]) is written to the UNet's per-row timestep input -- fusion_sampling.py:324-327,336 (`latent_model_input`, `t`)
__global__ void __launch_bounds__(256)
step_prologue_kernel(const float* __restrict__ x, float* __restrict__ latent, float* __restrict__ t_dev,
                     const float* __restrict__ prm, int rows, int64_t n) {
    const int64_t sd = blockIdx.y;
    const float4* src = (const float4*)(x + sd * n);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        for (int r = 0; r < rows; ++r) ((float4*)(latent + (sd * rows + r) * n))[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < rows) t_dev[sd * rows + threadIdx.x] = prm[0];
}

// ------------------------------------------------------------------ video sampler (I2VGen-XL loop, config #5)
template <int DT> struct StT;
template <> struct StT<TMIX_F32>  { static __device__ __forceinline__ void st(float* p, int64_t i, float v) { p[i] = v; } };
template <> struct StT<TMIX_F16>  { static __device__ __forceinline__ void st(__half* p, int64_t i, float v) { p[i] = __float2half_rn(v); } };
template <> struct StT<TMIX_BF16> { static __device__ __forceinline__ void st(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); } };

// video_gen/pipeline_i2vgen_xl.py:699-719 in one pass: CFG on the v-prediction, eps / x0 from (x, v), DDIM move.
// Latents and predictions share the model dtype in that loop, so in fp16 mode EVERY binary op rounds (rnd<DT>).
template <int DT>
__global__ void __launch_bounds__(256)
vpred_step_kernel(const typename EpsT<DT>::T* __restrict__ x, const typename EpsT<DT>::T* __restrict__ v,
                  typename EpsT<DT>::T* __restrict__ out, int64_t n, float g, float sa, float s1, float sa_n, float s1_n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float xv = EpsT<DT>::ld(x, i);
        const float vv = cfg<DT>(EpsT<DT>::ld(v, i), EpsT<DT>::ld(v, n + i), g);
        const float eps = rnd<DT>(rnd<DT>(sa * vv) + rnd<DT>(s1 * xv));
        const float x0 = rnd<DT>(rnd<DT>(sa * xv) - rnd<DT>(s1 * vv));
        StT<DT>::st(out, i, rnd<DT>(rnd<DT>(sa_n * x0) + rnd<DT>(s1_n * eps)));
    }
}

// video_gen/utils_attn.py:433-455: frames 1.. of every clip <- first frame (hard) or interp*first + (1-interp)*frame
template <int DT>
__global__ void __launch_bounds__(256)
frame_inject_kernel(typename EpsT<DT>::T* __restrict__ x, int64_t per_frame, int frames, int64_t n, int hard, float a, float b) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t per_clip = (int64_t)(frames - 1) * per_frame;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t clip = i / per_clip, r = i - clip * per_clip;
        const int64_t e = r % per_frame, base = clip * frames * per_frame;
        const float first = EpsT<DT>::ld(x, base + e);
        const int64_t dst = base + per_frame + r;
        const float o = hard ? first : rnd<DT>(rnd<DT>(a * first) + rnd<DT>(b * EpsT<DT>::ld(x, dst)));
        StT<DT>::st(x, dst, o);
    }
}

template <int DT>
int launch_vpred(const void* x, const void* v, void* out, int64_t n, float g, float sa, float s1, float sa_n, float s1_n, hipStream_t st) {
    typedef typename EpsT<DT>::T T;
    int64_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    vpred_step_kernel<DT><<<blocks, 256, 0, st>>>((const T*)x, (const T*)v, (T*)out, n, g, sa, s1, sa_n, s1_n);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

template <int DT>
int launch_inject(void* x, int64_t per_frame, int frames, int64_t n, int hard, float a, float b, hipStream_t st) {
    typedef typename EpsT<DT>::T T;
    int64_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    frame_inject_kernel<DT><<<blocks, 256, 0, st>>>((T*)x, per_frame, frames, n, hard, a, b);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

}  // namespace

extern "C" int tmix_vpred_step(const void* x, const void* v, void* out, int dtype, int64_t n, float g,
                               float sa, float s1, float sa_next, float s1_next, void* stream) {
    if (!x || !v || !out) TMIX_FAIL(TMIX_EINVAL, "vpred_step: null pointer");
    if (n < 1) TMIX_FAIL(TMIX_ESHAPE, "vpred_step: empty latent");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case TMIX_F32:  return launch_vpred<TMIX_F32>(x, v, out, n, g, sa, s1, sa_next, s1_next, st);
    case TMIX_F16:  return launch_vpred<TMIX_F16>(x, v, out, n, g, sa, s1, sa_next, s1_next, st);
    case TMIX_BF16: return launch_vpred<TMIX_BF16>(x, v, out, n, g, sa, s1, sa_next, s1_next, st);
    }
    TMIX_FAIL(TMIX_EINVAL, "vpred_step: bad dtype %d", dtype);
}

extern "C" int tmix_frame_inject(void* x, int dtype, int clips, int frames, int64_t per_frame, int hard, float interp,
                                 float one_minus_interp, void* stream) {
    if (!x) TMIX_FAIL(TMIX_EINVAL, "frame_inject: null pointer");
    if (clips < 1 || frames < 2 || per_frame < 1) TMIX_FAIL(TMIX_ESHAPE, "frame_inject: clips=%d frames=%d per_frame=%lld", clips, frames, (long long)per_frame);
    const int64_t n = (int64_t)clips * (frames - 1) * per_frame;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case TMIX_F32:  return launch_inject<TMIX_F32>(x, per_frame, frames, n, hard, interp, one_minus_interp, st);
    case TMIX_F16:  return launch_inject<TMIX_F16>(x, per_frame, frames, n, hard, interp, one_minus_interp, st);
    case TMIX_BF16: return launch_inject<TMIX_BF16>(x, per_frame, frames, n, hard, interp, one_minus_interp, st);
    }
    TMIX_FAIL(TMIX_EINVAL, "frame_inject: bad dtype %d", dtype);
}

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

extern "C" int tmix_fused_tweedie_step_dev(const float* x, const void* eps, int eps_dtype, const float* masks,
                                           int64_t mask_seed_stride, float* out_x, float* out_x0, int K, int channels,
                                           int64_t hw, int mode, int rows, int seeds, const float* params, void* stream) {
    if (!x || !eps || !out_x || !params) TMIX_FAIL(TMIX_EINVAL, "tweedie_step_dev: null pointer");
    if (mode < TMIX_STEP_FUSION || mode > TMIX_STEP_RESAMPLE) TMIX_FAIL(TMIX_EINVAL, "tweedie_step_dev: bad mode %d", mode);
    if (mode == TMIX_STEP_FUSION && (!masks || K < 1)) TMIX_FAIL(TMIX_EINVAL, "tweedie_step_dev: FUSION needs masks and K>=1");
    if (mode == TMIX_STEP_RESAMPLE && K < 1) TMIX_FAIL(TMIX_EINVAL, "tweedie_step_dev: RESAMPLE needs K>=1");
    if (channels < 1 || hw < 1 || seeds < 1 || seeds > 65535) TMIX_FAIL(TMIX_ESHAPE, "tweedie_step_dev: channels=%d hw=%lld seeds=%d", channels, (long long)hw, seeds);
    const int need = mode == TMIX_STEP_PLAIN ? 2 : K + 1;
    if (rows < need) TMIX_FAIL(TMIX_ESHAPE, "tweedie_step_dev: mode %d needs %d eps rows per seed, got %d", mode, need, rows);
    const int64_t n = (int64_t)channels * hw;
    hipStream_t st = (hipStream_t)stream;
    switch (eps_dtype) {
    case TMIX_F32:  return launch_dev<TMIX_F32>(x, eps, masks, mask_seed_stride, out_x, out_x0, K, n, hw, mode, rows, seeds, params, st);
    case TMIX_F16:  return launch_dev<TMIX_F16>(x, eps, masks, mask_seed_stride, out_x, out_x0, K, n, hw, mode, rows, seeds, params, st);
    case TMIX_BF16: return launch_dev<TMIX_BF16>(x, eps, masks, mask_seed_stride, out_x, out_x0, K, n, hw, mode, rows, seeds, params, st);
    }
    TMIX_FAIL(TMIX_EINVAL, "tweedie_step_dev: bad eps dtype %d", eps_dtype);
}

extern "C" int tmix_step_prologue(const float* x, float* latent, float* t_dev, const float* params, int seeds, int rows,
                                  int64_t n, void* stream) {
    if (!x || !latent || !t_dev || !params) TMIX_FAIL(TMIX_EINVAL, "step_prologue: null pointer");
    if (seeds < 1 || seeds > 65535 || rows < 1 || rows > 256 || n < 4 || (n & 3)) TMIX_FAIL(TMIX_ESHAPE, "step_prologue: seeds=%d rows=%d n=%lld (n %% 4 == 0)", seeds, rows, (long long)n);
    if (!aligned16(x) || !aligned16(latent)) TMIX_FAIL(TMIX_EALIGN, "step_prologue: x / latent must be 16-byte aligned");
    int64_t bx = (n / 4 + 255) / 256; if (bx > 256) bx = 256;
    step_prologue_kernel<<<dim3((unsigned)bx, (unsigned)seeds), 256, 0, (hipStream_t)stream>>>(x, latent, t_dev, params, rows, n);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}
