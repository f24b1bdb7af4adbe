// L2 -> LDS / L2 -> VGPR fill rate of one CU: how many bytes per clock do the load paths of the GEMM mainloop deliver?
// Every workgroup streams its own 64 KiB window (L2 resident, larger than the 32 KiB L1) over and over.
//   mode 0: buffer_load_dwordx4 ... lds   linear source rows (8 lanes x 16 B = one 128 B row, 8 rows per instruction)
//   mode 1: same, XOR-swizzled chunk order inside each row (the GEMM's source-side swizzle)
//   mode 2: global_load_dwordx4 into VGPRs, linear
//   mode 3: buffer_load_dword ... lds (4 B per lane)
//   mode 4: mode 0 with every lane's row 2 KiB apart (the GEMM's real pattern: 8 rows of a K-panel, row stride = K*2 bytes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void blds4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 4, voff, soff, 0, 0);
}

template <int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) k_dma(const char* src, float* out, int iters, int window) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * window;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, window, 0x00020000);
    constexpr int PER_IT = 32 * 1024;                 // bytes per iteration per workgroup
    constexpr int INSTR = PER_IT / 1024 / NW;         // dwordx4 instructions per wave per iteration
    float acc = 0.f;
    unsigned voff;
    if (MODE == 1) voff = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) * 16);
    else if (MODE == 4) voff = (lane >> 3) * 2048 + (lane & 7) * 16;
    else voff = lane * 16;
    for (int it = 0; it < iters; ++it) {
        const unsigned wbase = (unsigned)((it & 1) * PER_IT);
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < INSTR; ++i) {
                const float4 v = *(const float4*)(base + wbase + (i * NW + w) * 1024 + lane * 16);
                acc += v.x + v.w;
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < INSTR * 4; ++i) blds4(rs, lane * 4, wbase + (i * NW + w) * 256, smem + (i * NW + w) * 256);
        } else if (MODE == 4) {
            // 16 rows x 2 KiB window: instruction j covers 8 rows x 128 B at column block j
#pragma unroll
            for (int i = 0; i < INSTR; ++i) blds16(rs, voff, wbase + ((i * NW + w) & 15) * 128 + ((i * NW + w) >> 4) * 16384, smem + (i * NW + w) * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < INSTR; ++i) blds16(rs, voff, wbase + (i * NW + w) * 1024, smem + (i * NW + w) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (MODE != 2) acc = ((float*)smem)[tid];
    out[blockIdx.x * blockDim.x + tid] = acc;
}

template <int MODE, int NW>
void run(const char* name, const char* src, float* out, int wgs, int window) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_dma<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    k_dma<MODE, NW><<<wgs, NW * 64, 64 * 1024>>>(src, out, 50, window);
    hipEventRecord(e0);
    k_dma<MODE, NW><<<wgs, NW * 64, 64 * 1024>>>(src, out, iters, window);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * iters * 32 * 1024;
    const double per_cu = 32.0 * 1024 * iters * (wgs / 256.0) / (ms * 1e-3 * 2.4e9);
    printf("%-44s wgs=%4d waves/WG=%d: %7.3f ms  %7.2f TB/s  %6.1f B/clk/CU (at 2.4 GHz)\n", name, wgs, NW, ms, bytes / ms / 1e9, per_cu);
}

int main() {
    const int window = 64 * 1024;
    char* src; float* out;
    hipMalloc(&src, (size_t)1024 * window); hipMemset(src, 1, (size_t)1024 * window);
    hipMalloc(&out, 1024 * 512 * 4);
    for (int wgs : {256, 512}) {
        run<0, 4>("lds-dma x4 linear", src, out, wgs, window);
        run<1, 4>("lds-dma x4 xor-swizzled rows", src, out, wgs, window);
        run<4, 4>("lds-dma x4 rows 2 KiB apart", src, out, wgs, window);
        run<2, 4>("global_load x4 -> VGPR", src, out, wgs, window);
        run<3, 4>("lds-dma dword", src, out, wgs, window);
        run<0, 8>("lds-dma x4 linear", src, out, wgs, window);
        run<2, 8>("global_load x4 -> VGPR", src, out, wgs, window);
    }
    return 0;
}
