// A GEMM K loop without the arithmetic being right: what does a 128 x 160 x 64 K-tile (36 KiB of operands, 24 ds_read_b128 + 20 MFMAs per math wave) cost
// with the operands staged by (MODE 0) LDS-DMA or (MODE 1) global_load_dwordx4 -> VGPR -> ds_write_b128, issued by NL loader waves (NL = 0: by the four
// math waves themselves, between their MFMAs)?  BAR = 1: one s_barrier per K-tile over all waves, as in the real loop; 0: free-running.
// RD / MF: ds_read_b128 and MFMAs per math wave and K-tile (0 switches that part off).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, char* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int PIECES = 36;           // 1 KiB pieces per K-tile
constexpr int NSLOT = 4;             // ring slots of 36 KiB

// REAL = 1: the footprint of the real launch -- A [4096 x ld] and W [1280 x ld] row-major, workgroup -> (tile_m, tile_n) by the GEMM kernel's XCD-aware
// grouping (8 A tiles x 4 W tiles per XCD), K walked once per repetition (ld / 128 K-tiles), `reps` repetitions: the lab's "hot" state.
template <int MODE, int NL, int D, int BAR, int RD, int MF, int REAL = 0>
__global__ void __launch_bounds__((4 + NL) * 64) k_loop(const char* src, float* out, int ktiles, int ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int region = (blockIdx.x & 7) * 4 + ((blockIdx.x >> 3) & 3);
    const char* base = src + (size_t)region * 288 * ld;          // 288 operand rows (A 128 + W 160) of this workgroup, shared inside the XCD
    unsigned radd_a = 0, radd_w = 0;                              // REAL: row offsets of this workgroup's A / W tile inside one [4096 + 1280] x ld matrix
    if (REAL) {
        const int id = (blockIdx.x & 7) * 32 + (blockIdx.x >> 3), grp = id >> 6, rem = id & 63;
        const int tile_n = rem >> 3, tile_m = grp * 8 + (rem & 7);
        base = src;
        radd_a = (unsigned)(tile_m * 128) * (unsigned)ld;
        radd_w = (unsigned)(4096 + tile_n * 160 - 128) * (unsigned)ld;      // pieces 16.. are W rows (piece * 8 - 128)
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, REAL ? (4096 + 1280) * ld : 288 * ld, 0x00020000);
    const unsigned voff0 = (lane >> 3) * ld + (((lane & 7) ^ (lane >> 3)) * 16);
    auto pv = [&](int piece) { return voff0 + (unsigned)(piece * 8) * ld + (REAL ? (piece < 16 ? radd_a : radd_w) : 0u); };
    constexpr int NST = NL ? NL : 4;                 // staging waves
    constexpr int PPW = PIECES / NST;                // pieces per staging wave and K-tile
    const bool loader = NL && w >= 4;
    const int sid = NL ? w - 4 : w;
    float accs = 0.f;
    if (loader || NL == 0) {
        // ---- staging stream (loader waves; with NL = 0 it is interleaved into the math loop below instead)
    }
    if (loader) {
        float4 v[MODE == 1 ? D : 1];
        int issued = 0;
        if (MODE == 1) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int pc = d % PPW, kt = d / PPW;
                v[d] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, pv(pc * NST + sid), (unsigned)(kt * 128) % (unsigned)ld, 0));
            }
        }
        constexpr int G = MODE == 1 ? D / PPW : 1;       // K-tiles in flight (register sets)
        static_assert(MODE != 1 || D % PPW == 0, "D in whole K-tiles");
        for (int kt = 0; kt < ktiles; kt += G) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                char* slot = smem + ((kt + g) % NSLOT) * PIECES * 1024;
                if (MODE == 0) {
#pragma unroll
                    for (int pc = 0; pc < PPW; ++pc) blds16(rs, pv(pc * NST + sid), (unsigned)((kt + g) * 128) % (unsigned)ld, slot + (pc * NST + sid) * 1024);
                    wait_vmcnt<PPW * 2>();              // two K-tiles stay in flight
                } else if (MODE == 1) {
#pragma unroll
                    for (int pc = 0; pc < PPW; ++pc) {
                        wait_vmcnt<D - 1>();
                        *(float4*)(slot + (pc * NST + sid) * 1024 + lane * 16) = v[g * PPW + pc];
                        v[g * PPW + pc] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, pv(pc * NST + sid), (unsigned)((kt + g + G) * 128) % (unsigned)ld, 0));
                    }
                }
                if (BAR) __builtin_amdgcn_s_barrier();
            }
        }
        wait_vmcnt<0>();
        if (MODE == 1) accs = v[0].x;
        out[blockIdx.x * blockDim.x + tid] = accs;
        return;
    }
    // ---- math waves
    f32x16 c[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) c[j] = (f32x16){0};
    frag_ab fa = {0}, fb[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) fb[j] = (frag_ab){0};
    float4 v[(NL == 0 && MODE == 1) ? 9 : 1];
    for (int kt = 0; kt < ktiles; ++kt) {
        const char* slot = smem + (kt % NSLOT) * PIECES * 1024;
        char* nslot = smem + ((kt + 2) % NSLOT) * PIECES * 1024;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (MF) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa, c[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (RD) {
                    // one (and for j == 0 two) fragment reads per MFMA: 6 per k-step, 24 per K-tile
                    const frag_ab t = *(const frag_ab*)(slot + 16384 + ((j * 32 + (lane & 31)) * 128 + (((kk * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4)));
                    fb[j] = t;
                    if (j == 0) fa = *(const frag_ab*)(slot + ((w * 32 + (lane & 31)) * 128 + (((kk * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4)));
                }
                if (NL == 0) {
                    const int q = kk * 5 + j;
                    if (q < 9) {
                        if (MODE == 0) blds16(rs, pv(q * 4 + w), (unsigned)((kt + 2) * 128) % (unsigned)ld, nslot + (q * 4 + w) * 1024);
                        else {
                            *(float4*)(nslot + (q * 4 + w) * 1024 + lane * 16) = v[q];
                            v[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, pv(q * 4 + w), (unsigned)((kt + 3) * 128) % (unsigned)ld, 0));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (NL == 0 && MODE == 0) wait_vmcnt<9>();
        if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    wait_vmcnt<0>();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) s += c[j][0] + (float)fb[j][0];
    out[blockIdx.x * blockDim.x + tid] = s + (float)fa[0] + v[0].x;
}

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f);
        p[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}

template <int MODE, int NL, int D, int BAR, int RD, int MF, int REAL = 0>
void run(const char* name, const char* src, float* out, int ld) {
    const int ktiles = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int smem = NSLOT * PIECES * 1024;
    auto k = k_loop<MODE, NL, D, BAR, RD, MF, REAL>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    k<<<256, (4 + NL) * 64, smem>>>(src, out, 50, ld);
    hipEventRecord(e0);
    k<<<256, (4 + NL) * 64, smem>>>(src, out, ktiles, ld);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-26s %s ld=%5d loaders=%d D=%2d bar=%d reads=%d mfma=%d: %7.1f ns per K-tile  (%5.1f B/clk/CU@2.4GHz; MFMA alone = 267 ns @2.4GHz)\n", name, REAL ? "REAL" : "syn ", ld, NL, D, BAR, RD, MF,
           ms * 1e6 / ktiles, 36864.0 / (ms * 1e6 / ktiles * 2.4));
}

int main() {
    char* src; float* out;
    hipMalloc(&src, (size_t)256 * 1024 * 1024); hipMemset(src, 1, (size_t)256 * 1024 * 1024);
    hipMalloc(&out, 1024 * 1024 * 4);
    const int ld = 2560;
    // math part alone
    run<2, 4, 9, 0, 0, 1>("no staging: mfma", src, out, ld);
    run<2, 4, 9, 0, 1, 0>("no staging: reads", src, out, ld);
    run<2, 4, 9, 0, 1, 1>("no staging: reads + mfma", src, out, ld);
    run<2, 4, 9, 1, 1, 1>("no staging: reads + mfma", src, out, ld);
    run<0, 0, 9, 1, 1, 1>("lds-dma by math waves", src, out, ld);
    run<0, 0, 9, 1, 0, 1>("lds-dma by math waves", src, out, ld);
    run<0, 0, 9, 1, 1, 0>("lds-dma by math waves", src, out, ld);
    run<1, 0, 9, 1, 1, 1>("reg-staged by math waves", src, out, ld);
    run<1, 0, 9, 1, 0, 1>("reg-staged by math waves", src, out, ld);
    run<0, 4, 9, 1, 1, 1>("lds-dma loaders", src, out, ld);
    run<0, 4, 9, 0, 1, 1>("lds-dma loaders", src, out, ld);
    run<0, 2, 18, 1, 1, 1>("lds-dma loaders", src, out, ld);
    run<1, 4, 9, 1, 1, 1>("reg-staged loaders", src, out, ld);
    run<1, 4, 18, 1, 1, 1>("reg-staged loaders", src, out, ld);
    run<1, 4, 27, 1, 1, 1>("reg-staged loaders", src, out, ld);
    run<1, 4, 18, 0, 1, 1>("reg-staged loaders", src, out, ld);
    run<1, 2, 18, 1, 1, 1>("reg-staged loaders", src, out, ld);
    run<1, 2, 36, 1, 1, 1>("reg-staged loaders", src, out, ld);
    run<1, 1, 36, 1, 1, 1>("reg-staged loaders", src, out, ld);
    run<1, 4, 18, 1, 0, 1>("reg-staged loaders", src, out, ld);
    run<1, 4, 18, 1, 1, 0>("reg-staged loaders", src, out, ld);
    run<1, 4, 18, 1, 0, 0>("reg-staged loaders", src, out, ld);
    run<0, 4, 9, 1, 0, 0>("lds-dma loaders", src, out, ld);
    // the real launch's footprint (A 4096 x K, W 1280 x K, XCD-grouped tiles), first on the memset operands, then on random bf16 in [-1, 1)
    for (int rnd = 0; rnd < 2; ++rnd) {
        if (rnd) { fill_bf16<<<2048, 256>>>((unsigned short*)src, (size_t)128 * 1024 * 1024, 7u); hipDeviceSynchronize(); printf("-- random operands\n"); }
        for (int l : {10240, 2560}) {
            run<0, 4, 9, 1, 1, 1, 1>("lds-dma loaders", src, out, l);
            run<0, 4, 9, 1, 0, 1, 1>("lds-dma loaders", src, out, l);
            run<0, 4, 9, 1, 1, 0, 1>("lds-dma loaders", src, out, l);
            run<0, 4, 9, 1, 0, 0, 1>("lds-dma loaders", src, out, l);
            run<0, 0, 9, 1, 1, 1, 1>("lds-dma by math waves", src, out, l);
            run<1, 4, 27, 1, 1, 1, 1>("reg-staged loaders", src, out, l);
        }
        run<0, 4, 9, 1, 1, 1, 0>("lds-dma loaders", src, out, 2560);
        run<2, 4, 9, 1, 1, 1, 0>("no staging: reads + mfma", src, out, 2560);
    }
    return 0;
}
