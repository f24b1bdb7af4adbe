// micro-benchmark: cycles per wave64 instruction for a few VALU ops on gfx950 (one wave per SIMD, dependent-free streams)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 256
#define BODY(name, code) \
__global__ void k_##name(float* out, unsigned long long* cyc) { \
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = 0.999f, c = 0.001f; \
    unsigned long long t0 = __builtin_readcyclecounter(); \
    _Pragma("unroll 1") for (int i = 0; i < REP; ++i) { code } \
    unsigned long long t1 = __builtin_readcyclecounter(); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0; \
}
#define X8(op) op(a0) op(a1) op(a2) op(a3) op(a4) op(a5) op(a6) op(a7)
#define FMA(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(c));
#define MUL(v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(b));
#define EXP(v) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
#define MAX3(v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(c));
#define CVT(v) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v) : "v"(b));
#define PL32(v) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(b));
BODY(fma, X8(FMA))
BODY(mul, X8(MUL))
BODY(exp, X8(EXP))
BODY(max3, X8(MAX3))
BODY(cvt, X8(CVT))
BODY(pl32, X8(PL32))
__global__ void k_pkfma(float* out, unsigned long long* cyc) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {threadIdx.x * 0.001f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {0.999f, 0.998f}, c = {0.001f, 0.002f};
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < REP; ++i) {
#define PKFMA(v) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(c));
        X8(PKFMA)
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K> void run(const char* name, K k, int waves_per_simd) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    int threads = 256 * waves_per_simd;     // one block on one CU: threads/64 waves spread over 4 SIMDs
    k<<<1, threads>>>(out, cyc); hipDeviceSynchronize();
    k<<<1, threads>>>(out, cyc); hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-8s waves/SIMD=%d: %.2f s_memtime ticks per instruction (per wave)\n", name, waves_per_simd, (double)h / (REP * 8));
    hipFree(out); hipFree(cyc);
}
__global__ void k_spin(unsigned long long n, unsigned long long* out) {
    unsigned long long t0 = __builtin_readcyclecounter(), t1 = t0;
    while (t1 - t0 < n) t1 = __builtin_readcyclecounter();
    out[0] = t1 - t0;
}
__global__ void k_spin2(unsigned long long n, unsigned long long* out) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1 = t0;
    while (t1 - t0 < n) t1 = __builtin_amdgcn_s_memtime();
    out[0] = t1 - t0;
}
int main() {
    { unsigned long long* o; hipMalloc(&o, 8); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      for (int which = 0; which < 2; ++which) {
        hipEventRecord(a);
        if (which == 0) k_spin<<<1, 64>>>(10000000ull, o); else k_spin2<<<1, 64>>>(10000000ull, o);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s: 1e7 ticks took %.3f ms -> %.1f MHz\n", which ? "s_memtime" : "readcyclecounter", ms, 1e7 / ms / 1e3); } }

    for (int w = 1; w <= 4; w *= 2) {
        run("fma", k_fma, w); run("mul", k_mul, w); run("pk_fma", k_pkfma, w); run("exp", k_exp, w);
        run("max3", k_max3, w); run("cvt_pk", k_cvt, w); run("pl32swap", k_pl32, w);
    }
    return 0;
}
