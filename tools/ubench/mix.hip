// does a wave's VALU stream overlap with another wave's MFMA stream on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 frag;
typedef __attribute__((ext_vector_type(4))) float f4;
template <int NM, int NV>
__global__ void k_mix(float* out, unsigned long long* cyc) {
    frag a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(i * 0.5f); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    float kb = 0.999f, kc = 0.001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int m = 0; m < NM / 4; ++m) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < NV / 8; ++v) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(kb), "v"(kc));
            F(v0) F(v1) F(v2) F(v3) F(v4) F(v5) F(v6) F(v7)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if (threadIdx.x == blockDim.x - 64 && blockIdx.x == 0) cyc[0] = t1 - t0;   // last wave of the block
}
template <typename K> void run(const char* name, K k, int wps) {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<256, 256 * wps>>>(out, cyc); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<<<256, 256 * wps>>>(out, cyc);
    hipEventRecord(b); hipError_t e = hipEventSynchronize(b);
    if (e != hipSuccess || hipGetLastError() != hipSuccess) printf("ERR\n");
    float ms; hipEventElapsedTime(&ms, a, b);
    double cycles = ms / 5 * 1e-3 * 2.4e9;     // wall cycles per launch (64 iterations per wave)
    printf("%-22s waves/SIMD=%d: %.0f wall cycles per iteration per SIMD -> %.0f per wave-iteration\n", name, wps, cycles / 64, cycles / 64 / wps);
}
int main() {
    for (int w = 1; w <= 4; ++w) {
        run("32 MFMA only", k_mix<32, 0>, w);
        run("160 VALU only", k_mix<0, 160>, w);
        run("32 MFMA + 160 VALU", k_mix<32, 160>, w);
    }
}
