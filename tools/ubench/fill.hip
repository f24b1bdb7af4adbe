// THROUGHPUT of the operand paths into a CU (dma.hip measures one vmcnt(0) + barrier round trip per 32 KiB, i.e. latency):
// every wave keeps D 1-KiB requests in flight (counted vmcnt, no barrier, wave-private LDS slots) and streams an L2-resident window.
//   path 0: buffer_load_dwordx4 ... lds (LDS-DMA)       path 1: global_load_dwordx4 -> VGPR
//   path 2: both, alternating (D/2 of each in flight)    path 3: LDS-DMA from waves 0..NW/2-1, ds_read_b128 streams from the others
//   path 4: LDS-DMA + MFMAs in the same wave (32x32x16 bf16, one per request)
// window layouts: own = every workgroup its own 64 KiB; xcd = the 32 workgroups of an XCD share one 256 KiB window (GEMM panels)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int PATH, int NW, int D>
__global__ void __launch_bounds__(NW * 64) k_fill(const char* src, float* out, int iters, int window, int shared_xcd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wg = shared_xcd ? (blockIdx.x & 7) : blockIdx.x;
    const char* base = src + (size_t)wg * window;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, window, 0x00020000);
    char* slot = smem + w * D * 1024;
    const unsigned voff = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) * 16);
    const unsigned mask = (unsigned)window - 1;
    // a different phase per wave and workgroup so that sharers do not walk in lock step
    unsigned pos = (unsigned)((w * 37 + blockIdx.x * 11) * 1024) & mask;
    float acc = 0.f;
    float4 v[PATH == 1 || PATH == 2 ? D : 1];
    f32x16 c = {0};
    frag_ab fa = {0}, fb = {0};
    if (PATH == 3 && w >= NW / 2) {
        // reader waves: ds_read_b128 stream over the whole LDS block
        const int n = iters * 4;
        float4 s = make_float4(0, 0, 0, 0);
        for (int it = 0; it < n; it += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 t = *(const float4*)(smem + (((it + u) * 1024 + lane * 16) & (NW * D * 1024 - 1)));
                s.x += t.x; s.y += t.w;
            }
        }
        out[blockIdx.x * blockDim.x + tid] = s.x + s.y;
        return;
    }
    // prologue: D requests
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (PATH == 1 || (PATH == 2 && (d & 1))) v[d] = *(const float4*)(base + pos + lane * 16);
        else blds16(rs, voff, pos, slot + d * 1024);
        pos = (pos + NW * 1024) & mask;
    }
    for (int it = 0; it < iters; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            wait_vmcnt<D - 1>();
            if (PATH == 1 || (PATH == 2 && (d & 1))) { acc += v[d].x; v[d] = *(const float4*)(base + pos + lane * 16); }
            else blds16(rs, voff, pos, slot + d * 1024);
            if (PATH == 4) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c, 0, 0, 0);
            pos = (pos + NW * 1024) & mask;
        }
    }
    wait_vmcnt<0>();
    if (PATH == 1 || PATH == 2) {
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d].y;
    }
    acc += ((float*)smem)[tid] + c[0];
    out[blockIdx.x * blockDim.x + tid] = acc;
}


// The MFMA-operand gather: a wave reads ITS 32 (16) rows of a row-major K-panel straight into fragment registers -- lane (row, half) takes 16 bytes,
// so one instruction touches 32 (16) different 128-byte lines.  G = 32: v_mfma_32x32x16 layout (2 lanes per row, 4 instructions per 128-byte K-tile),
// G = 16: v_mfma_16x16x32 layout (4 lanes per row, 2 instructions per row block and K-tile, two row blocks).  MIX = 1: plus five LDS-DMA pieces per
// K-tile and wave (the W tile of a 128 x 160 GEMM tile) -- the proportions of the A-direct mainloop.  MIX = 2: the same A bytes loaded COALESCED (8 lanes per
// row) and parked in wave-private LDS with ds_write_b128 (transpose through LDS), plus the five pieces.
template <int G, int MIX, int NW, int D>
__global__ void __launch_bounds__(NW * 64) k_gather(const char* src, const char* wsrc, float* out, int ktiles, int ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int region = (blockIdx.x & 7) * 4 + ((blockIdx.x >> 3) & 3);      // 4 A tiles per XCD, each shared by 8 workgroups
    const char* base = src + (size_t)region * 128 * ld + (size_t)w * 32 * ld;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 32 * ld, 0x00020000);
    const char* wbase = wsrc + (size_t)(blockIdx.x & 7) * 8 * 160 * ld + (size_t)((blockIdx.x >> 5) & 7) * 160 * ld;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 160 * ld, 0x00020000);
    unsigned voff;
    if (MIX == 2) voff = (lane >> 3) * ld + (lane & 7) * 16;
    else if (G == 32) voff = (lane & 31) * ld + (lane >> 5) * 16;
    else voff = (lane & 15) * ld + (lane >> 4) * 16;
    const unsigned wvoff = (lane >> 3) * ld + (((lane & 7) ^ (lane >> 3)) * 16);
    char* slot = smem + w * (D * 5 * 1024 + 4096);
    float4 v[D][4];
    float acc = 0.f;
    auto issue = [&](int d, int kt) __attribute__((always_inline)) {
        const unsigned so = (unsigned)(kt * 128) % (unsigned)ld;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned off;
            if (MIX == 2) off = q * 8 * ld;                         // 8 rows per instruction, whole 128-byte lines
            else if (G == 32) off = q * 32;                           // k-step q: chunks 2q, 2q+1
            else off = (q & 1) * 64 + (q >> 1) * 16 * ld;           // k-step q&1 of row block q>>1
            v[d][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + off, so, 0));
        }
        if (MIX) {
#pragma unroll
            for (int q = 0; q < 5; ++q) blds16(rw, wvoff + (unsigned)((q * NW + w) * 8) * ld, so, slot + (d * 5 + q) * 1024);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    for (int kt = D; kt < ktiles; kt += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            wait_vmcnt<(D - 1) * (MIX ? 9 : 4)>();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MIX == 2) *(float4*)(slot + D * 5 * 1024 + q * 1024 + lane * 16) = v[d][q];
                else acc += v[d][q].x;
            }
            issue(d, kt + d);
        }
    }
    wait_vmcnt<0>();
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += v[d][q].y;
    acc += ((float*)smem)[tid];
    out[blockIdx.x * blockDim.x + tid] = acc;
}

template <int G, int MIX, int NW, int D>
void run_gather(const char* name, const char* src, float* out, int ld) {
    const int ktiles = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int smem = NW * (D * 5 * 1024 + 4096);
    hipFuncSetAttribute((const void*)k_gather<G, MIX, NW, D>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const char* wsrc = src + (size_t)64 * 1024 * 1024;
    k_gather<G, MIX, NW, D><<<256, NW * 64, smem>>>(src, wsrc, out, 64, ld);
    hipEventRecord(e0);
    k_gather<G, MIX, NW, D><<<256, NW * 64, smem>>>(src, wsrc, out, ktiles, ld);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_kt = (MIX ? 9.0 : 4.0) * 1024 * NW;
    const double bytes = 256.0 * ktiles * per_kt;
    printf("%-44s ld=%5d waves=%d D=%d: %7.3f ms %7.2f TB/s %6.1f B/clk/CU@2.4GHz  %6.1f ns per K-tile (%.0f KiB)\n", name, ld, NW, D, ms, bytes / ms / 1e9,
           bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e6 / ktiles, per_kt / 1024);
}

template <int PATH, int NW, int D>
void run(const char* name, const char* src, float* out, int wgs, int window, int shared) {
    const int iters = 4096;      // (4 MiB per wave: a 512 KiB window is walked several times)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int smem = NW * D * 1024;
    hipFuncSetAttribute((const void*)k_fill<PATH, NW, D>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    k_fill<PATH, NW, D><<<wgs, NW * 64, smem>>>(src, out, 64, window, shared);
    hipEventRecord(e0);
    k_fill<PATH, NW, D><<<wgs, NW * 64, smem>>>(src, out, iters, window, shared);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int loaders = PATH == 3 ? NW / 2 : NW;
    const double bytes = (double)wgs * loaders * (iters + D) * 1024;
    printf("%-34s %s wgs=%4d waves=%d D=%2d (%3d KiB in flight/WG): %7.3f ms %7.2f TB/s %6.1f B/clk/CU@2.4GHz\n", name, shared ? "xcd" : "own", wgs, NW, D,
           loaders * D, ms, bytes / ms / 1e9, bytes / (wgs > 256 ? 256 : wgs) / (ms * 1e-3 * 2.4e9));
}

int main() {
    char* src; float* out;
    hipMalloc(&src, (size_t)1024 * 256 * 1024); hipMemset(src, 1, (size_t)1024 * 256 * 1024);
    hipMalloc(&out, 2048 * 1024 * 4);
    for (int shared = 0; shared < 2; ++shared) {
        const int window = shared ? 256 * 1024 : 64 * 1024;
        run<0, 4, 4>("lds-dma", src, out, 256, window, shared);
        run<0, 4, 8>("lds-dma", src, out, 256, window, shared);
        run<0, 4, 16>("lds-dma", src, out, 256, window, shared);
        run<0, 4, 32>("lds-dma", src, out, 256, window, shared);
        run<0, 8, 16>("lds-dma", src, out, 256, window, shared);
        run<0, 2, 32>("lds-dma", src, out, 256, window, shared);
        run<0, 1, 32>("lds-dma", src, out, 256, window, shared);
        run<1, 4, 8>("global->vgpr", src, out, 256, window, shared);
        run<1, 4, 16>("global->vgpr", src, out, 256, window, shared);
        run<1, 4, 32>("global->vgpr", src, out, 256, window, shared);
        run<1, 8, 16>("global->vgpr", src, out, 256, window, shared);
        run<2, 4, 16>("both alternating", src, out, 256, window, shared);
        run<2, 4, 32>("both alternating", src, out, 256, window, shared);
        run<3, 8, 16>("lds-dma + 4 ds_read waves", src, out, 256, window, shared);
        run<4, 4, 16>("lds-dma + mfma same wave", src, out, 256, window, shared);
        run<4, 4, 32>("lds-dma + mfma same wave", src, out, 256, window, shared);
    }
    // beyond the L2s: 128 MB in total (Infinity Cache resident after the warm-up pass), nothing shared -- what the fabric delivers to 256 streaming CUs
    run<0, 8, 16>("lds-dma, 512 KiB windows (MALL)", src, out, 256, 512 * 1024, 0);
    run<1, 8, 16>("global->vgpr, 512 KiB windows (MALL)", src, out, 256, 512 * 1024, 0);
    run<0, 4, 32>("lds-dma, 512 KiB windows (MALL)", src, out, 256, 512 * 1024, 0);
    run<0, 4, 32>("lds-dma, 1 MiB windows (HBM+MALL)", src, out, 256, 1024 * 1024, 0);
    // fewer CUs active: is the limit per CU or per chip / XCD?
    run<0, 4, 32>("lds-dma, 64 WGs", src, out, 64, 64 * 1024, 0);
    run<0, 4, 32>("lds-dma, 128 WGs", src, out, 128, 64 * 1024, 0);
    run<0, 4, 32>("lds-dma, 8 WGs", src, out, 8, 64 * 1024, 0);
    run<1, 4, 32>("global->vgpr, 64 WGs", src, out, 64, 64 * 1024, 0);

    for (int ld : {2560, 10240}) {
        run_gather<32, 0, 4, 2>("A gather 32x32x16 layout", src, out, ld);
        run_gather<32, 0, 4, 4>("A gather 32x32x16 layout", src, out, ld);
        run_gather<16, 0, 4, 4>("A gather 16x16x32 layout", src, out, ld);
        run_gather<32, 2, 4, 4>("A coalesced -> VGPR -> ds_write + 5 W dma", src, out, ld);
        run_gather<32, 1, 4, 2>("A gather 32x32x16 + 5 W lds-dma", src, out, ld);
        run_gather<32, 1, 4, 4>("A gather 32x32x16 + 5 W lds-dma", src, out, ld);
        run_gather<16, 1, 4, 4>("A gather 16x16x32 + 5 W lds-dma", src, out, ld);
    }
    return 0;
}
