// how long does a wave spend ISSUING LDS-DMA instructions (cycles it cannot issue MFMAs in), and how long waiting?
// t0 | N x buffer_load_dwordx4..lds | t1 | s_waitcnt vmcnt(0) | t2 ; one workgroup per CU, NW waves, clock = s_memtime
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
typedef __attribute__((ext_vector_type(8))) __bf16 frag;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int N, int NW, int MFMA>
__global__ void __launch_bounds__(NW * 64) k(const char* src, unsigned long long* cyc, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536, 0x00020000);
    unsigned long long issue = 0, wait = 0, mf = 0;
    frag a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.01f); b[i] = (__bf16)(i * 0.5f); }
    f16v c = {0};
    for (int it = 0; it < iters; ++it) {
        const unsigned wb = (it & 1) * 32768;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < N; ++i) blds16(rs, lane * 16, wb + ((i * NW + w) & 31) * 1024, smem + ((i * NW + w) & 31) * 1024);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (MFMA) {
#pragma unroll
            for (int m = 0; m < MFMA; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        }
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        issue += t1 - t0; mf += t2 - t1; wait += t3 - t2;
    }
    if (lane == 0) { cyc[(blockIdx.x * NW + w) * 3] = issue; cyc[(blockIdx.x * NW + w) * 3 + 1] = mf; cyc[(blockIdx.x * NW + w) * 3 + 2] = wait; }
    out[blockIdx.x * blockDim.x + tid] = c[0] + ((float*)smem)[tid];
}

// interleaved variant: the same wave issues G MFMAs after every DMA instruction (does the matrix pipe keep working on the
// MFMAs already issued while the wave is blocked in the next DMA issue?)
template <int N, int NW, int G>
__global__ void __launch_bounds__(NW * 64) ki(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536, 0x00020000);
    frag a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.01f); b[i] = (__bf16)(i * 0.5f); }
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        const unsigned wb = (it & 1) * 32768;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            blds16(rs, lane * 16, wb + ((i * NW + w) & 31) * 1024, smem + ((i * NW + w) & 31) * 1024);
#pragma unroll
            for (int m = 0; m < G; m += 4) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * blockDim.x + tid] = c0[0] + c1[0] + c2[0] + c3[0] + ((float*)smem)[tid];
}
template <int N, int NW, int G> void runi(const char* src, float* out) {
    const int iters = 1000, wgs = 256;
    hipFuncSetAttribute((const void*)ki<N, NW, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    ki<N, NW, G><<<wgs, NW * 64, 64 * 1024>>>(src, out, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    ki<N, NW, G><<<wgs, NW * 64, 64 * 1024>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("interleaved: %d waves x %2d x [1 DMA + %d MFMA]: kernel %.3f us/iter\n", NW, N, G, ms * 1e3 / iters);
}

// classic path: global_load_dwordx4 into VGPRs interleaved with MFMAs, ds_write_b128 after the wait (what LDS-DMA replaces)
template <int N, int NW, int G>
__global__ void __launch_bounds__(NW * 64) kg(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * 65536;
    frag a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.01f); b[i] = (__bf16)(i * 0.5f); }
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        const unsigned wb = (it & 1) * 32768;
        float4 v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            v[i] = *(const float4*)(base + wb + ((i * NW + w) & 31) * 1024 + lane * 16);
#pragma unroll
            for (int m = 0; m < G; m += 4) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) *(float4*)(smem + ((i * NW + w) & 31) * 1024 + lane * 16) = v[i];
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + tid] = c0[0] + c1[0] + c2[0] + c3[0] + ((float*)smem)[tid];
}
template <int N, int NW, int G> void rung(const char* src, float* out) {
    const int iters = 1000, wgs = 256;
    hipFuncSetAttribute((const void*)kg<N, NW, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    kg<N, NW, G><<<wgs, NW * 64, 64 * 1024>>>(src, out, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kg<N, NW, G><<<wgs, NW * 64, 64 * 1024>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("global_load->VGPR->ds_write: %d waves x %2d x [1 load + %d MFMA]: kernel %.3f us/iter\n", NW, N, G, ms * 1e3 / iters);
}

// warp-specialised variant: waves [0, NL) only issue DMA (N each), waves [NL, NW) only issue MFMAs
template <int N, int NL, int NW, int MFMA>
__global__ void __launch_bounds__(NW * 64) ks(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536, 0x00020000);
    frag a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.01f); b[i] = (__bf16)(i * 0.5f); }
    f16v c = {0};
    for (int it = 0; it < iters; ++it) {
        const unsigned wb = (it & 1) * 32768;
        if (w < NL) {
#pragma unroll
            for (int i = 0; i < N; ++i) blds16(rs, lane * 16, wb + ((i * NL + w) & 31) * 1024, smem + ((i * NL + w) & 31) * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int m = 0; m < MFMA; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * blockDim.x + tid] = c[0] + ((float*)smem)[tid];
}
template <int N, int NL, int NW, int MFMA> void runs(const char* src, float* out) {
    const int iters = 1000, wgs = 256;
    hipFuncSetAttribute((const void*)ks<N, NL, NW, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    ks<N, NL, NW, MFMA><<<wgs, NW * 64, 64 * 1024>>>(src, out, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    ks<N, NL, NW, MFMA><<<wgs, NW * 64, 64 * 1024>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("specialised: %d loader waves x %2d DMA + %d math waves x %2d MFMA: kernel %.3f us/iter\n", NL, N, NW - NL, MFMA, ms * 1e3 / iters);
}

template <int N, int NW, int MFMA> void run(const char* src, unsigned long long* cyc, float* out) {
    const int iters = 1000, wgs = 256;
    hipFuncSetAttribute((const void*)k<N, NW, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    k<N, NW, MFMA><<<wgs, NW * 64, 64 * 1024>>>(src, cyc, out, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<N, NW, MFMA><<<wgs, NW * 64, 64 * 1024>>>(src, cyc, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3 * 8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on this part: report ns per iteration as well as the kernel-time-derived figure
    printf("N=%2d DMA/wave, %d waves, %2d MFMA: kernel %.3f us/iter | memtime ticks/iter: issue %.2f mfma %.2f wait %.2f\n", N, NW, MFMA,
           ms * 1e3 / iters, (double)h[0] / iters, (double)h[1] / iters, (double)h[2] / iters);
}

int main() {
    char* src; unsigned long long* cyc; float* out;
    hipMalloc(&src, (size_t)256 * 65536); hipMemset(src, 1, (size_t)256 * 65536);
    hipMalloc(&cyc, 256 * 8 * 3 * 8); hipMalloc(&out, 256 * 512 * 4);
    run<1, 4, 0>(src, cyc, out); run<2, 4, 0>(src, cyc, out); run<4, 4, 0>(src, cyc, out); run<8, 4, 0>(src, cyc, out); run<16, 4, 0>(src, cyc, out);
    run<8, 4, 16>(src, cyc, out); run<8, 4, 32>(src, cyc, out); run<0, 4, 16>(src, cyc, out); run<0, 4, 32>(src, cyc, out);
    run<4, 8, 0>(src, cyc, out); run<4, 8, 16>(src, cyc, out);
    runi<8, 4, 4>(src, out); runi<8, 4, 2 * 2>(src, out); runi<4, 8, 4>(src, out); runi<8, 4, 8>(src, out); runi<6, 8, 4>(src, out);
    rung<8, 4, 4>(src, out); rung<8, 4, 0>(src, out); rung<4, 8, 4>(src, out); rung<8, 4, 8>(src, out); rung<6, 8, 4>(src, out); rung<4, 8, 0>(src, out);
    runs<8, 4, 8, 16>(src, out); runs<8, 4, 8, 32>(src, out); runs<8, 4, 8, 0>(src, out); runs<0, 4, 8, 32>(src, out);
    runs<16, 2, 6, 32>(src, out); runs<32, 1, 5, 32>(src, out); runs<8, 4, 12, 16>(src, out); runs<9, 4, 8, 20>(src, out);
    return 0;
}
