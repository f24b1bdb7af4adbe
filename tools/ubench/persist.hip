// persist.hip -- what would a PERSISTENT transformer-block kernel pay between two dependent stages, against the kernel boundary it replaces?
//
// A chain of S dependent "stages" on 256 workgroups (one per CU: 144 KB of LDS requested), each stage = what a GEMM launch of the captured step does
// at its edges: read R bytes that OTHER workgroups (other XCDs) wrote in the previous stage, spin T ns (the launch's body), write W bytes of output.
//   mode 0: S kernel launches captured into ONE hipGraph on one stream (today's step: boundary = drain + L2 write-back + dispatch + cold first loads)
//   mode 1: one persistent launch, stages separated by a grid barrier: release (write-back of the stage's dirty lines) -> arrive on ONE device-scope counter
//           -> spin on it (s_sleep) -> acquire (L2 / L1 invalidate)
//   mode 2: the same with a two-level arrive: per-XCD counter (32 workgroups), the XCD's last arriver bumps the global counter everybody polls
// Both chains compute the same function of the data, and the result buffers are compared: a barrier that lets a stale line through fails the check.
// Output: per-stage time minus T = the cost of the stage's edges.    ./persist [S] [T_ns] [R_KB] [W_KB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NWG = 256, NT = 256;

__device__ __forceinline__ void body(const uint4* __restrict__ in, uint4* __restrict__ out, int s, int wg, int T_ticks, int r_vec, int w_vec, int nprod) {
    // read r_vec 16-byte vectors spread over nprod producers' chunks (chunk = w_vec vectors), fold them, spin, write this workgroup's chunk
    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
    const int per = r_vec / nprod;
    for (int j = 0; j < nprod; ++j) {
        const int src = (wg * 7 + j * 37 + 1) & (NWG - 1);
        const uint4* p = in + (size_t)src * w_vec;
        for (int i = threadIdx.x; i < per; i += NT) { const uint4 v = p[i % w_vec]; acc.x += v.x; acc.y ^= v.y; acc.z += v.z * 3u; acc.w ^= v.w + (unsigned)j; }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < T_ticks) __builtin_amdgcn_s_sleep(2);
    uint4* q = out + (size_t)wg * w_vec;
    for (int i = threadIdx.x; i < w_vec; i += NT) q[i] = make_uint4(acc.x + (unsigned)(i + s), acc.y ^ (unsigned)(i * 2654435761u), acc.z + (unsigned)wg, acc.w ^ (unsigned)s);
}

__global__ void __launch_bounds__(NT) stage_kernel(const uint4* in, uint4* out, int s, int T_ticks, int r_vec, int w_vec, int nprod) {
    body(in, out, s, blockIdx.x, T_ticks, r_vec, w_vec, nprod);
}

__global__ void __launch_bounds__(NT) persist_kernel(uint4* b0, uint4* b1, int S, int T_ticks, int r_vec, int w_vec, int nprod, unsigned* ctr, int two_level,
                                                       unsigned long long* tstamp) {
    const int wg = blockIdx.x;
    unsigned xcc = 0;
    if (two_level) xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;     // HW_REG_XCC_ID[3:0]
    unsigned* gctr = ctr;                 // global counter
    unsigned* xctr = ctr + 32 + xcc * 32; // per-XCD counters, one cache line apart
    for (int s = 0; s < S; ++s) {
        body((s & 1) ? b1 : b0, (s & 1) ? b0 : b1, s, wg, T_ticks, r_vec, w_vec, nprod);
        if (s + 1 == S) break;
        __syncthreads();                  // the workgroup's stores are issued
        if (threadIdx.x == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            if (two_level) {
                const unsigned old = __hip_atomic_fetch_add(xctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if ((old & 31) == 31) __hip_atomic_fetch_add(gctr, 32u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // (32 workgroups per XCD)
            } else __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(s + 1) * NWG;
            // (relaxed polls: an ACQUIRE load per poll invalidates the caches every time round -- 27 us per barrier in the first version of this file)
            while (__hip_atomic_load(gctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            if (tstamp && wg == 0) tstamp[s] = __builtin_amdgcn_s_memrealtime() - t0;
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);     // every wave drops its stale lines (the scope of the fence is the agent)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 64;
    const int T_ns = argc > 2 ? atoi(argv[2]) : 15000;
    const int R_KB = argc > 3 ? atoi(argv[3]) : 128;
    const int W_KB = argc > 4 ? atoi(argv[4]) : 40;
    const int r_vec = R_KB * 64, w_vec = W_KB * 64, nprod = 8, T_ticks = T_ns / 10;
    const size_t bytes = (size_t)NWG * w_vec * 16;
    uint4 *a0, *a1, *p0, *p1; unsigned* ctr; unsigned long long* ts;
    CK(hipMalloc(&a0, bytes)); CK(hipMalloc(&a1, bytes)); CK(hipMalloc(&p0, bytes)); CK(hipMalloc(&p1, bytes));
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&ts, 8 * 1024));
    std::vector<unsigned> init(bytes / 4);
    for (size_t i = 0; i < init.size(); ++i) init[i] = (unsigned)(i * 2654435761u) >> 7;
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(hipFuncSetAttribute((const void*)stage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    CK(hipFuncSetAttribute((const void*)persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // mode 0: S launches in one graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < S; ++s) stage_kernel<<<NWG, NT, 144 * 1024, st>>>((s & 1) ? a1 : a0, (s & 1) ? a0 : a1, s, T_ticks, r_vec, w_vec, nprod);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best[3] = {1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemcpy(a0, init.data(), bytes, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best[0]) best[0] = ms;
    }
    std::vector<unsigned> ref(bytes / 4), got(bytes / 4);
    CK(hipMemcpy(ref.data(), (S & 1) ? a1 : a0, bytes, hipMemcpyDeviceToHost));
    printf("S=%d stages, body %d ns, reads %d KB from %d other workgroups, writes %d KB per workgroup, %d workgroups\n", S, T_ns, R_KB, nprod, W_KB, NWG);
    printf("mode 0  graph of %d launches      : %8.1f us total, %6.2f us per stage, edges %6.2f us per stage\n", S, best[0] * 1e3, best[0] * 1e3 / S, best[0] * 1e3 / S - T_ns * 1e-3);
    for (int mode = 1; mode <= 2; ++mode) {
        int bad = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemcpy(p0, init.data(), bytes, hipMemcpyHostToDevice));
            CK(hipMemsetAsync(ctr, 0, 4096, st));
            CK(hipEventRecord(e0, st));
            persist_kernel<<<NWG, NT, 144 * 1024, st>>>(p0, p1, S, T_ticks, r_vec, w_vec, nprod, ctr, mode == 2, ts);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best[mode]) best[mode] = ms;
        }
        CK(hipMemcpy(got.data(), (S & 1) ? p1 : p0, bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
        std::vector<unsigned long long> t(S);
        CK(hipMemcpy(t.data(), ts, 8 * (S - 1), hipMemcpyDeviceToHost));
        double sum = 0; for (int s = 0; s + 1 < S; ++s) sum += (double)t[s];
        printf("mode %d  persistent, %s barrier: %8.1f us total, %6.2f us per stage, edges %6.2f us per stage; workgroup 0 waits %5.2f us in the barrier on average; result %s\n",
               mode, mode == 2 ? "two-level" : "one-level", best[mode] * 1e3, best[mode] * 1e3 / S, best[mode] * 1e3 / S - T_ns * 1e-3, sum / (S - 1) * 0.01, bad ? "DIFFERS" : "identical");
    }
    return 0;
}
