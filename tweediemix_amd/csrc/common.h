// common.h -- shared helpers for the gfx950 kernels of libtmix_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/tmix.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short  bf16x8;   // one MFMA A/B fragment (4 VGPR)
typedef __attribute__((ext_vector_type(4))) float  f32x4;    // 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 accumulator fragment

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even, on gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the integer form -- NaN test, rounding
// add, shift, merge -- was ~14 VALU per pair, and with one wave per SIMD the GEMM epilogues are VALU-issue bound: tools/jobs/r3za_epi_abl.sh)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
#if defined(TMIX_ABL_SW_BF16)   // dev A/B builds only (tools/build_variant.sh): the integer form this replaced
    uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
    a = ((a & 0x7fffffffu) > 0x7f800000u) ? ((a >> 16) | 0x40u) : ((a + 0x7fffu + ((a >> 16) & 1u)) >> 16);
    b = ((b & 0x7fffffffu) > 0x7f800000u) ? ((b >> 16) | 0x40u) : ((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
    return (a & 0xffffu) | (b << 16);
#else
    const f32x2_hw f = {lo, hi};
    const bf16x2_hw b = __builtin_convertvector(f, bf16x2_hw);
    return __builtin_bit_cast(uint32_t, b);
#endif
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, f) & 0xffffu); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (diffusers GEGLU uses F.gelu with approximate='none'): gelu(x) = x * Phi(x).  Phi(-a) = exp2(P(a)) with P the degree-7
// polynomial fit of log2 Phi(-a) on a in [0, 5.5] (Chebyshev nodes; tools/fit_gelu.py), Phi(x) = x < 0 ? Phi(-|x|) : 1 - Phi(-|x|):
// ONE transcendental (v_exp_f32) and 7 fma.  |gelu error| <= 7e-7 absolute and <= 5e-6 RELATIVE (also deep in the negative tail, where
// the Abramowitz-Stegun 7.1.26 erf used before -- a v_rcp_f32 plus a v_exp_f32, quarter-rate instructions both -- had 1.7e-3 relative);
// the GEGLU epilogue of FF1 is VALU-bound and spent a third of its instructions here.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float a = fminf(fabsf(x), 5.5f);
    float p = -1.921234625e-06f;
    p = __builtin_fmaf(p, a, 6.328849712e-05f);
    p = __builtin_fmaf(p, a, -9.434208502e-04f);
    p = __builtin_fmaf(p, a, 8.556272268e-03f);
    p = __builtin_fmaf(p, a, -5.405228293e-02f);
    p = __builtin_fmaf(p, a, -4.583817849e-01f);
    p = __builtin_fmaf(p, a, -1.151278331e+00f);
    p = __builtin_fmaf(p, a, -9.999938561e-01f);
    const float e = __builtin_amdgcn_exp2f(p);                 // Phi(-|x|)
    return x * (x < 0.f ? e : 1.0f - e);                        // (0.5 + copysign(0.5 - e, x) would cancel in the tail)
}

// smallest E8M0 exponent e (value 2^e, stored as e + 127) with amax * 2^-e <= 448, the e4m3 maximum; 0 for an all-zero block
__device__ __forceinline__ int e8m0_for_amax(float amax) {
    if (!(amax > 0.f)) return 0;
    const unsigned bits = __float_as_uint(amax * (1.0f / 448.0f));
    int e = (int)((bits >> 23) & 0xff) - 127 + ((bits & 0x7fffff) ? 1 : 0);
    return e < -127 ? -127 : (e > 127 ? 127 : e);
}
// 2^-e, e in [-127, 127] (e = 127 -- only reachable from non-finite input -- is the denormal 2^-127, not 0: 0 * inf would be NaN bytes)
__device__ __forceinline__ float exp2_neg_int(int e) { return e >= 127 ? __uint_as_float(0x00400000u) : __uint_as_float((unsigned)(127 - e) << 23); }

// thread-local error string (host side)
void tmix_set_error(const char* fmt, ...);

// The library's environment switches (debug / A-B knobs).  Read ONCE per process, on first use: a launch, the *_launches / *_ws_bytes query that accounts for it and a
// captured graph must all see the same value, and an eager launch does not scan the environment.  tmix_env_refresh() re-reads them (tests flip a switch mid-process).
enum { TMIX_ENV_GN_NO_SMALL = 0, TMIX_ENV_ATTN_NO_SPLIT, TMIX_ENV_ATTN_GENERAL, TMIX_ENV_NARROW_EPILOGUE, TMIX_ENV_COUNT };
bool tmix_env(int which);
#define TMIX_FAIL(code, ...) do { tmix_set_error(__VA_ARGS__); return (code); } while (0)
#define TMIX_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    tmix_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return (int)e_; } } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- in-situ launch timing (tmix_prof_begin / tmix_prof_end): while a host thread has profiling switched on, every
// instrumented launch (GEMM, conv, attention, GroupNorm) takes the next 8-word slot of the caller's device buffer --
// also when the launch is being captured into a hipGraph, so replays of that graph time the launches in the schedule the
// product really runs (two concurrent chains, graph dependencies), which rocprofv3 serialises.  Slot layout, ticks of the
// 100 MHz s_memrealtime clock: {start, end, sum(t1 - t0), sum(t2 - t0), sum(t3 - t0), workgroups, -, -}.
// Plain mode costs a launch next to nothing: workgroup 0 (the first one dispatched) stores the start, the LAST 32
// workgroups of the grid (dispatch is in order, so the last finisher is among them) store their end time with a
// write-through store and the last one to land wins (within a store latency of the true maximum; read-modify-write
// atomics -- or even plain stores -- on one address from thousands of workgroups stretched 4096-workgroup launches).  Detail mode (tools/gemm_lab) adds the per-workgroup phase sums with atomics: t0 = entry, t1 = first operands
// landed, t2 = main loop done, t3 = epilogue stores issued; start / end become exact min / max.
struct TmixProf { unsigned long long* buf; int cap, next, detail; };
TmixProf& tmix_prof_state();
static inline unsigned long long* tmix_prof_take(int* detail = nullptr) {
    TmixProf& st = tmix_prof_state();
    if (detail) *detail = st.detail;
    if (!st.buf || st.next >= st.cap) return nullptr;
    return st.buf + 8 * (size_t)(st.next++);
}
// tmix_gemm_prefetch_next: the hint for the next GEMM / conv launch of this thread (capi.cpp); taking it clears it
void tmix_prefetch_take(const char** ptr, long long* bytes);
__device__ __forceinline__ unsigned long long prof_now() { return __builtin_amdgcn_s_memrealtime(); }
// call from ONE thread per workgroup; `first` = this is the workgroup dispatched first (linear block id 0)
__device__ __forceinline__ unsigned long long prof_enter(unsigned long long* slot, bool first, int detail) {
    const unsigned long long t = prof_now();
    if (detail) atomicMin(slot, t);
    else if (first) __hip_atomic_store(slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
}
__device__ __forceinline__ void prof_leave(unsigned long long* slot, int detail, unsigned long long t0, unsigned long long t1, unsigned long long t2) {
    if (detail) {
        const unsigned long long t3 = prof_now();
        atomicMax(slot + 1, t3);
        atomicAdd(slot + 2, t1 - t0); atomicAdd(slot + 3, t2 - t0); atomicAdd(slot + 4, t3 - t0); atomicAdd(slot + 5, 1ull);
    } else {
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y * gridDim.z;
        if (lin + 32 >= total) __hip_atomic_store(slot + 1, prof_now(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Touch every 64-byte line of the kernel-argument block at kernel entry.  A launch's arguments are cold in the scalar cache, and the compiler loads
// the fields of a large by-value parameter struct where it first needs them -- one s_load + s_waitcnt round trip after the other down the prologue
// (the GEMM kernel: ~20 of them in front of the first LDS-DMA request).  With all lines requested back to back the misses overlap and the later
// loads hit.  BYTES = sizeof(the kernel's parameter struct).
template <int BYTES> __device__ __forceinline__ void kernarg_touch() {
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    constexpr int LINES = (BYTES + 63) / 64;
    static_assert(LINES >= 1 && LINES <= 8, "kernarg block larger than expected");
    // ONE statement: the loads and the wait for them (scalar loads return asynchronously -- the compiler must not reuse a destination before then)
    unsigned t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile("s_load_dword %0, %8, 0x0\n\t"
                 "s_load_dword %1, %8, %9\n\t"
                 "s_load_dword %2, %8, %10\n\t"
                 "s_load_dword %3, %8, %11\n\t"
                 "s_load_dword %4, %8, %12\n\t"
                 "s_load_dword %5, %8, %13\n\t"
                 "s_load_dword %6, %8, %14\n\t"
                 "s_load_dword %7, %8, %15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
                 : "s"(ka), "n"(LINES > 1 ? 0x40 : 0), "n"(LINES > 2 ? 0x80 : 0), "n"(LINES > 3 ? 0xc0 : 0), "n"(LINES > 4 ? 0x100 : 0),
                   "n"(LINES > 5 ? 0x140 : 0), "n"(LINES > 6 ? 0x180 : 0), "n"(LINES > 7 ? 0x1c0 : 0)
                 : "memory");
}

// XCD-aware bijective remap of a linear workgroup id (guide T1): consecutive logical ids land on
// the same XCD (hardware places physical id b on XCD b % 8).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
// The same over a (tiles, slices) grid: workgroups are dispatched x-fastest and dealt to the XCDs round robin by their LINEAR id, so a remap of blockIdx.x
// alone gives every XCD a few tiles of EVERY slice -- with one weight set per slice (concept-routed LoRA rows) each XCD then pulls all the sets and every
// slice's A rows through the fabric (routed 1280-cube: 55 MB of operand traffic per launch chip-wide).  Remapping the linear id over the whole grid gives an XCD
// one contiguous run of (slice, tile) ids: half a slice per XCD at B = 4 (34 MB).  bid = tile id inside the slice, by = slice.
__device__ __forceinline__ void xcd_remap_grid(int& bid, int& by) {
    const int gx = gridDim.x, gy = gridDim.y;
    if (gy == 1) { bid = xcd_remap(blockIdx.x, gx); by = blockIdx.y; return; }
    const int lg = xcd_remap(blockIdx.y * gx + blockIdx.x, gx * gy);
    by = lg / gx; bid = lg - by * gx;
}
