// gemm_kernel.h -- bf16 MFMA GEMM (C = A * W^T) and 3x3 NHWC implicit-GEMM convolution for gfx950.
//
// One templated mainloop <BM, BN, WM x WN waves, NS stages> serves both:
//   * operands are staged HBM->LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into an
//     NS-deep ring of [rows][64] bf16 tiles; the loop keeps NS-2 whole K-tiles in flight across the
//     per-tile barrier with COUNTED s_waitcnt vmcnt(N) + raw s_barrier (never a draining __syncthreads),
//     because these GEMMs are latency-bound, not MFMA-bound, with a one-tile prefetch distance;
//   * the LDS image is XOR-swizzled: 16-byte chunk c of row r lives at chunk position c ^ ((r>>1)&7).
//     LDS-DMA destinations are lane-linear, so the permutation is applied to each lane's SOURCE address
//     and undone on the ds_read_b128 side; it is conflict-free for the 16-lane service groups of
//     ds_read_b128 with 32-row MFMA fragments;
//   * v_mfma_f32_32x32x16_bf16, wave tile (BM/WM) x (BN/WN), fp32 accumulation;
//   * the convolution differs only in how a lane finds its source address (im2col on the fly: the K-tiles walk
//     channel-chunk major -- 64 channels of the shifted pixel, the nine taps of a chunk back to back; padding taps read
//     zeros via the buffer bounds check); stride-2 and nearest-x2 upsampling are folded into the gather;
//   * fused epilogues: bias, per-row-group bias (time embedding), residual add, GEGLU, transposed store
//     (V^T for the attention kernel), per-batch weight sets (concept routing).
//
// MFMA roofline: 2*M*N*K flops per launch against the 2.5 PFLOP/s dense bf16 peak.
#pragma once
#include "common.h"
#include <type_traits>
#include <stdlib.h>

// The kernel template is instantiated in gemm_inst_*.hip (one group of tilings per translation unit, built in parallel);
// gemm_conv.hip holds the host entry points and the tiling switch.
namespace tmix_gemm {

// TMIX_ABL (dev builds under tools/ab/ only; the shipped library is built without it): ablations that locate the bound of a
// launch -- bit 0: every workgroup stages tile (0, 0) (operands L2-hot, no fabric traffic), bit 1: no MFMAs (fragments are read
// and kept alive), bit 2: no LDS-DMA inside the K loop (the prologue's tiles are re-read), bit 3: no epilogue at all; staged plain epilogue only:
// bit 4: no bias loads, bit 5: no C stores (everything else runs), bit 6: the residual is neither requested nor added.
#ifndef TMIX_ABL
#define TMIX_ABL 0
#endif
constexpr int ABL = TMIX_ABL;

constexpr int BK = 64;
typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;

static __device__ __forceinline__ float xor32_sum(float x) {      // x + (value of lane ^ 32)
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// maximum over the four lanes of a quad (the MX block of an e4m3 copy is the 4 adjacent lanes of a row) on DPP quad_perm moves:
// two VALU instructions instead of two ds_bpermute round trips
static __device__ __forceinline__ float quad_max(float x) {
    float y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));      // lane ^ 1
    x = fmaxf(x, y);
    y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));            // lane ^ 2
    return fmaxf(x, y);
}

// DPP move within rows of 16 lanes (CTRL 0x120 + n: row_ror:n -- lane l reads lane (l - n) mod 16 of its row)
template <int CTRL> static __device__ __forceinline__ float dpp_row(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// two reductions for the price of one: a summed over the lane pairs (l, l ^ 32) lands in lanes 0-31, b in lanes 32-63 ...
static __device__ __forceinline__ float swap32_sum(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// ... and over (l, l ^ 16): a lands in the even rows of 16 lanes, b in the odd ones
static __device__ __forceinline__ float swap16_sum(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct Params {
    const bf16_t* A; int64_t lda, strideA;
    const bf16_t* W; int64_t ldw, strideW;
    bf16_t* C; int64_t ldc, strideC;
    const float* bias; int64_t strideBias;
    const bf16_t* R; int64_t ldr, strideR;
    const float* rgb; int rows_per_group;
    bf16_t* Ct; int64_t ldct, strideCt; int n_trans_begin;
    int M, N, K, tiles_m, tiles_n, group_m, epilogue;
    unsigned bytesA, bytesW;          // extents of one batch slice (buffer bounds)
    float* stats_out; int64_t strideStatsOut, ldStatsOut;      // LayerNorm producer side: float2 [tiles_n][ld] partial sums
    const float* ln_stats; int64_t strideLnStats, ldLnStats; const float* ln_colsum; int64_t strideLnColsum;   // consumer side
    float ln_inv_c, ln_eps; int ln_parts;
    // convolution geometry (CONV only)
    int H, Wd, Cin, Ho, Wo, mode, ntaps;
    unsigned long long* prof; int prof_detail;   // in-situ timing slot (common.h) or NULL
    int wide;                         // bit 0 / 1 / 2: the C / GEGLU / Ct stores may use the LDS-staged 16-byte form
    const unsigned char* scaleA; const unsigned char* scaleW; int64_t strideScaleA, strideScaleW;   // fp8: E8M0 exponent per A row / W row
    int64_t ldScaleA;                 // fp8 with MX block scales on A (PH = 3): scaleA is [K/32][ldScaleA] (k-block major)
    unsigned char* scale_out; int64_t ldScaleOut; int f8out;    // GEGLU output as e4m3 bytes + [N/64][ldScaleOut] block scales
    unsigned char* f8copy; int64_t ldF8copy;                    // plain epilogue: e4m3 COPY of the stored bf16 rows (+ scale_out [N/32][ldScaleOut])
    const char* pf; long long pf_bytes;                         // tmix_gemm_prefetch_next: the next launch's weights, touched in the prologue
    int pf_per;                                                 // 128-byte lines per touching thread (host-computed: a 64-bit division in every wave's prologue otherwise)
    int w_period, w_groups; unsigned w_magic;                   // > 0: batch slice b takes weight set (bias, ln_colsum, fp8 W scales) b % w_period; w_groups = batch / w_period, w_magic = floor(2^32 / w_groups) + 1
    float* cs_out;                                              // GroupNorm producer side: fp32 [M/32][2][N] column {sums | sums of squares} per 32-row block
    // convolution with 1x1 SHORTCUT taps (SC): behind the nine 3x3 taps the K loop walks the channels of up to two more NHWC tensors at the output pixel
    const bf16_t* S1; const bf16_t* S2; int c1s, c2s; unsigned bytesS1, bytesS2;
};

// LDS-DMA through a buffer descriptor: buffer_load_dwordx4 voff, rsrc, soff offen lds.  The per-lane part of the
// address is ONE 32-bit VGPR that stays constant across K-tiles (the K advance rides in the scalar soffset), and
// out-of-range offsets read as zero, which is how convolution padding taps are produced (no zero page, no selects).
static __device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char* lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

// MX block scales of A (PH = 3) live in LDS for the whole K loop: the K/32 blocks of a launch must fit beside the staging ring
constexpr int f8_block_cap(int bn) { return bn >= 256 ? 88 : 224; }

template <int N> static __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LW = 1 adds a LOADER wave to the WM x WN math waves (wave specialisation): measured on this chip (tools/ubench/dma_issue),
// an LDS-DMA instruction blocks the issuing wave for ~100 cycles and a wave's own MFMAs queue up behind its DMA issue
// (8 DMA + 32 MFMA per wave: 0.96 us per round), whereas DMA issued by one wave overlaps the MFMAs of OTHER waves almost
// perfectly (the same work split as 1 loader + 4 math waves: 0.53 us).  So with LW the math waves never touch VMEM in the
// K loop: the loader streams every K-tile (all (BM+BN)/8 instructions), waits for it with a counted vmcnt, and the
// per-iteration s_barrier hands it over.
// PH = 1 selects the PHASE-OFFSET mainloop (8 waves, two per SIMD): the K loop advances in 32-wide slices through a
// four-slot LDS ring (two slices stay in flight across every barrier), and each slice is a LOAD segment (LDS-DMA issue for
// slice s+3, fragment reads of slice s) followed by an MFMA segment (16 x v_mfma_f32_32x32x16_bf16 on a 128x64 wave tile at
// raised priority).  The second wave of every SIMD (waves 4-7) runs one barrier behind the first, so on each SIMD one wave
// issues DMA / ds_read while its partner keeps the matrix pipe busy -- an LDS-DMA instruction blocks its OWN wave's issue
// for ~100 cycles (tools/ubench/dma_issue), which is what held the lock-step loop below at ~50 % of the MFMA rate.
// KS = 2 (in-workgroup split-K): a SECOND group of WM x WN waves shares the tile; both groups stage (eight waves issue the
// LDS-DMA of a K-tile instead of four), group g multiplies k-steps 2g, 2g+1 of every 64-deep K-tile into its own accumulators,
// and after the loop group 1 hands its partial tile to group 0 through the (now free) staging ring.  For this path's
// one-tile-per-CU launches (4096 x 1280: 256 tiles of 128 x 160) the K loop is bound by how fast a CU can pull operands
// L2 -> LDS, and that rate grows with the number of waves issuing DMA (measured 38 GB/s per CU with four, 52 with eight).
// CS = 1: the instantiation that also leaves GroupNorm column statistics (Params::cs_out).  Its own kernel, not one more flavour inside the
// common one: with the two extra epilogue bodies compiled into every kernel the launches that do NOT use them lost 0.3 ms per step (same-box
// A/B, tools/jobs/r3zy_gn_fused.sh: GEMM class 19.21 -> 19.40 ms; the register allocation of the 256 x 320 tiling moved) -- as much as the
// statistics kernels they replace cost.  The CS = 0 kernels are the code they were before.
// EK: the epilogue family compiled into the instantiation -- 0: all of them (transposed regions, the narrow forms, everything below), 1: the staged GEGLU
// epilogue only, 2: the staged plain epilogue only (all its flavours), 3: the staged plain and the staged transposed form (q | k | V^T).  One kernel that can do everything carries every path's register demand: the
// 256 x 320 tiling compiled with all families spills 35 VGPRs (144 bytes of scratch per lane, 179 KB of code), with the GEGLU family alone none
// (227 VGPRs, 41 KB).  launch_cs picks the family from the launch's parameters; results are bit-identical (the same code, less of it).
// SC = 1 (convolution only): the 1x1 shortcut of a ResnetBlock2D rides in the same K loop -- after the nine taps of conv2 the loop walks the channels of the
// block's INPUT (one tensor, or the two halves of an up-block's concatenation, which then never has to be materialised) against weight rows
// [conv2 | conv_shortcut]: out = conv2(h) + conv_shortcut(x) from one accumulator, no shortcut GEMM, no residual round trip, no concat launch.
template <int BM, int BN, int WM, int WN, int NS, int CONV, int LW = 0, int PH = 0, int KS = 1, int CS = 0, int EK = 0, int SC = 0>
// (HIP's second launch-bounds argument is the minimum number of waves per SIMD: a workgroup with a loader wave puts three
// waves on one SIMD -- 2 x 5 or 1 x 9 waves per CU -- so those variants must fit 512/3 registers)
__global__ void __launch_bounds__((WM * WN * KS + LW) * 64, LW ? (NS * (BM + BN) * 128 > 80 * 1024 ? 2 : 3) : (KS == 1 && WM * WN == 4 && NS * (BM + BN) * 128 > 80 * 1024) ? 1 : 2)
gemm_conv_kernel(const Params p) {
    // PH = 4 / 5: fp8 operands (per-row / MX-block A scales, as PH = 2 / 3) in the LOCK-STEP loops below -- every non-phase tiling, loader waves included:
    // a staged row is still 128 bytes, i.e. 128 K values = TWO v_mfma_scale_f32_32x32x64_f8f6f4 k-steps per K-tile instead of four bf16 ones.  The loops
    // of this path are bound by operand bytes (tools/ubench/loop.hip with the launch's real footprint: 0.41 us per 36 KB K-tile for the memory system
    // alone, 0.45 us for MFMAs + fragment reads alone, 0.56 us together) -- e4m3 halves both.  MX-block A scales (PH = 5) ride in the ring: one more
    // LDS-DMA piece per K-tile, 4 blocks x BM rows of E8M0 bytes behind the W tile of the stage.
    constexpr bool PHL = PH >= 1 && PH <= 3;           // the phase-offset loop
    // (tile widths that the eight waves' 16-row LDS-DMA pieces do not divide -- 256 x 320: 20 W pieces -- re-stage the last piece in the surplus slots, like the
    // lock-step tilings: every wave issues the same count, so the counted waits stay uniform)
    static_assert(!PHL || (WM * WN == 8 && !LW && !CONV && NS == 4 && BM % 128 == 0 && BN % 32 == 0 && (BM / WM) % 32 == 0 && (BN / WN) % 32 == 0), "phase-offset mainloop geometry");
    static_assert(PH <= 3 || KS == 1, "fp8 lock-step loop: no in-workgroup split-K");
    // the convolution on e4m3 operands (PH = 5 only): the input is what a GroupNorm-apply pass wrote as e4m3 with MX block scales in the ROW-major form
    // [pixels][Cin / 32] (a tap shift then moves a row's scales by a multiple of 4 bytes); a K-tile is 128 channels of one tap, and each staging wave gathers the 4 scale
    // bytes of its BM / SW rows with one 4-byte LDS-DMA instruction
    static_assert(!(CONV && PH >= 4) || (PH == 5 && !SC && BM % ((LW ? LW : WM * WN)) == 0 && BM / (LW ? LW : WM * WN) <= 64), "fp8 convolution geometry");
    static_assert(KS == 1 || (KS == 2 && !LW && !PH && !CONV), "in-workgroup split-K geometry");
    static_assert(!SC || (CONV && (LW == 0 || LW == 2)), "shortcut taps belong to the convolution (lock-step tilings, with or without two loader waves)");
    // PH = 2: the same loop on OCP fp8 (e4m3) operands: a slice row is still 64 bytes, i.e. 64 K values, and the eight
    // v_mfma_scale_f32_32x32x64_f8f6f4 of a slice do the work of thirty-two bf16 MFMAs in the time of sixteen; every A row and every
    // W row carries ONE power-of-two scale (E8M0 byte) that the instruction applies itself -- constant along K, so a lane loads its
    // scales once and the K assignment inside a 64-byte slice need only be the same for both operands.
    // tilings that can also leave the e4m3 copy of C (register budget): no loader waves -- except the fp8 lock-step kernels, whose W fragments are
    // single-buffered (ROLL) to make room: every N = 1280 GEMM of an fp8 plan writes that copy, and it is the loader waves that make those launches fast
    // (as their own epilogue family EK = 4 -- the straight-line staged plain form WITH the copy and nothing else: inside the common family the loader-wave
    // kernel, at the 256-register limit of two waves per SIMD, spilled 5 dwords)
    constexpr bool F8C = !CONV && (LW ? (PH >= 4 && EK == 4) : true) && (BM / WM / 32) * (BN / WN / 32) <= 8;
    constexpr bool F8 = PH >= 2;
    constexpr bool F8B = PH == 3 || PH == 5;           // A carries one scale per 32 K values (MX blocks), streamed with the slices
    constexpr bool F8L = PH >= 4;                      // fp8 in the lock-step loops
    constexpr int SCP = (F8L && F8B) ? 1 : 0;          // one scale piece (1 KB slot, 4 * BM bytes used) per stage
    constexpr int EB = F8 ? 1 : 2;                     // bytes per operand element
    constexpr int NW = WM * WN;                        // math waves
    constexpr int TM = BM / WM, TN = BN / WN;          // wave tile
    constexpr int FM = TM / 32, FN = TN / 32;          // 32x32 fragments per wave
    constexpr int A_TILE = BM * 128, B_TILE = BN * 128, STAGE = A_TILE + B_TILE + SCP * 1024;
    static_assert(!SCP || 4 * BM <= 1024, "scale piece");
    constexpr int SLOT = (BM + BN) * 64;               // PH: one 32-wide K slice of both operands (rows of 64 bytes)
    constexpr int RING = PHL ? 4 * SLOT : NS * STAGE;   // bytes of the staging ring (the fused-LayerNorm block sits behind it)
    constexpr int SW = LW ? LW : NW * KS;              // waves that share the staging of a K-tile (LW: the loader waves)
    constexpr int IA = BM / 8, IB = BN / 8;            // LDS-DMA instructions per stage (8 rows of 128 bytes each)
    // ... per staging wave; when the waves do not divide them (64x160 over 5 waves) a surplus slot re-stages the last rows
    // (same bytes to the same place), so every wave issues the same count and the counted vmcnt waits stay valid
    constexpr int RA = (IA + SW - 1) / SW, RB = (IB + SW - 1) / SW;
    constexpr int L = RA + RB + SCP;                   // (every staging wave issues its share of the scale piece as one exec-masked instruction)
    static_assert(RA >= 1 && RB >= 1 && FM >= 1 && FN >= 1, "tile/wave geometry");
    static_assert(PHL || (NS - 2) * L <= 63, "vmcnt immediate");

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef TMIX_NO_KERNARG_TOUCH
    kernarg_touch<(int)sizeof(Params)>();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = KS > 1 ? w / NW : 0;                // split-K group of this wave
    const int wm_ = KS > 1 ? w - kg * NW : w;          // position among the group's math waves
    const int wr = wm_ / WN, wc = wm_ - wr * WN;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, (blockIdx.x | blockIdx.y) == 0, p.prof_detail);
    // ---- the NEXT launch's weights (tmix_gemm_prefetch_next): this workgroup's share, one dword per 128-byte line, requested
    // before the first operands of its own tile -- the prologue waits one memory round trip for those anyway, and these loads are
    // older in the queue, so the counted vmcnt waits below cover them.  The destinations stay reserved (pf_keep) until then.
    constexpr int PFU = 8;
    unsigned pf_keep[PFU];
#pragma unroll
    for (int u = 0; u < PFU; ++u) pf_keep[u] = 0;
    // (with loader waves only they touch: the math waves never wait on vmcnt, so nothing would cover their loads)
    if (p.pf && (!LW || w >= WM * WN)) {
        const long long nwg = (long long)gridDim.x * gridDim.y, nth = (LW ? LW : WM * WN * KS) * 64;
        const long long lines = (p.pf_bytes + 127) >> 7; const int per = p.pf_per;     // lines per thread
        const long long first = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * nth + (LW ? tid - WM * WN * 64 : tid);
#pragma unroll
        for (int u = 0; u < PFU; ++u) {
            const long long ln = first + (long long)u * nwg * nth;
            if (u < per && ln < lines) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_keep[u]) : "v"(p.pf + (ln << 7)) : "memory");
        }
    }
    const bool loader = LW && (w >= NW);               // wave-uniform role
    const bool stager = LW ? loader : true;
    const int sw_id = LW ? max(w - NW, 0) : w;         // this wave's slot among the staging waves

    // Tile order: each XCD (private 4 MiB L2) owns a contiguous range of logical ids, and ids sweep GM tile-rows
    // per tile-column, so the ~64 tiles resident on an XCD at any time form a compact GM x (64/GM) patch that
    // shares GM A-panels and 64/GM W-panels instead of streaming one W-panel per tile through the L2.
#ifdef TMIX_XCD_X_ONLY     // dev A/B builds: the remap of blockIdx.x alone (every XCD works on every slice)
    const int bid = xcd_remap(blockIdx.x, gridDim.x), by = blockIdx.y;
#else
    int bid, by;
    xcd_remap_grid(bid, by);
#endif
    const int per_group = p.group_m * p.tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * p.group_m;
    const int gsize = min(p.tiles_m - first_m, p.group_m);
    const int rem = bid - grp * per_group;
    const int tile_n = rem / gsize, tile_m = first_m + (rem - tile_n * gsize);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int m0l = (ABL & 1) ? 0 : m0, n0l = (ABL & 1) ? 0 : n0;      // rows the STAGING reads (ablation bit 0: tile (0, 0))
    // periodic weight sets (co-batched seeds: rows [seed][concept] share the concept's weights): slices that read the same W are issued back to back
    // (by / w_groups as a multiply-high by the host's reciprocal: scalar instructions only, exact for by, w_groups < 65536)
    const int bzw = p.w_period > 0 ? (int)__umulhi((unsigned)by, p.w_magic) : by;       // == bz % w_period
    const int bz = p.w_period > 0 ? (by - bzw * p.w_groups) * p.w_period + bzw : by;

    const bf16_t* Ab = (const bf16_t*)((const char*)p.A + (int64_t)bz * p.strideA * EB);      // strides count elements (fp8: bytes)
    const bf16_t* Wb = (const bf16_t*)((const char*)p.W + (int64_t)bzw * p.strideW * EB);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, p.bytesW, 0x00020000);

    // ---- per-lane staging sources. Wave-instruction idx = r*NW + w covers LDS rows idx*8 .. idx*8+7.
    const int lrow = lane >> 3;
    // per-lane offsets are 32-bit element counts off wave-uniform bases (saddr + voffset form of global_load_lds)
    // A loader wave (LW) covers ALL rows of the tile; its per-lane offsets follow from two parity variants (the swizzle
    // of instruction idx depends on idx & 1 only) plus a wave-uniform row advance, clamped like the per-wave arrays below.
    auto slot_a = [&](int r) { int i = r * SW + sw_id; if constexpr (IA % SW != 0) i = min(i, IA - 1); return i; };
    auto slot_w = [&](int r) { int i = r * SW + sw_id; if constexpr (IB % SW != 0) i = min(i, IB - 1); return i; };
    constexpr int RAa = (LW && !CONV) ? 1 : RA, RBa = LW ? 1 : RB;
    unsigned woff[RBa], aoff[RAa];
    int asw[(CONV && !LW) ? RA : 1];
    // convolution, per staged row: cbase = byte offset of the row's tap-(0, 0) source pixel (+ the row's swizzle; may be "negative": the tap delta brings it back),
    // cmask = bit t: tap t reads inside the image (else zeros through the bounds check); bits 16 / 17: parity of the row's upsampled tap-(0, 0) coordinates (UP2);
    // cpix (shortcut taps only) = the output pixel's index.  The K loop walks the taps of a channel chunk back to back, so a tap's offsets are derived every K-tile:
    // one add and one select per row on top of a scalar tap delta (recomputing them from (image, y, x) cost the 256 x 320 tiling 50 % of its time)
    unsigned cbase[CONV ? RA : 1], cmask[CONV ? RA : 1];
    int cpix[(CONV && SC) ? RA : 1];
    auto conv_row = [&](int r, int m, unsigned swb) __attribute__((always_inline)) {
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int py = rem / p.Wo, px = rem - py * p.Wo;
            int iy0, ix0; unsigned par = 0;
            if (p.mode == TMIX_CONV_S1 || p.mode == TMIX_CONV_T3) { iy0 = py - 1; ix0 = px - 1; }
            else if (p.mode == TMIX_CONV_S2) { iy0 = 2 * py - 1; ix0 = 2 * px - 1; }
            else if (p.mode == TMIX_CONV_S2A) { iy0 = 2 * py; ix0 = 2 * px; }
            else { iy0 = (py - 1) >> 1; ix0 = (px - 1) >> 1; par = (unsigned)((py - 1) & 1) << 16 | (unsigned)((px - 1) & 1) << 17; }     // nearest x2: source = upsampled coordinate >> 1
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t >= p.ntaps) break;
                const int ky = p.mode == TMIX_CONV_T3 ? t : t / 3, kx = p.mode == TMIX_CONV_T3 ? 1 : t - (t / 3) * 3;
                bool ok;
                if (p.mode == TMIX_CONV_UP2) { const int uy = py + ky - 1, ux = px + kx - 1; ok = (uy >= 0) & (uy < 2 * p.H) & (ux >= 0) & (ux < 2 * p.Wd); }
                else { const int iy = iy0 + ky, ix = ix0 + kx; ok = (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd); }
                mk |= (unsigned)ok << t;
            }
            cbase[r] = (unsigned)(((b * p.H + iy0) * p.Wd + ix0) * p.Cin) * (unsigned)EB + swb;
            cmask[r] = mk | par;
            if constexpr (SC) cpix[r] = (b * p.H + py) * p.Wd + px;
        }
    };
    unsigned aoffp[2], amaxp[2], woffp[2], wmaxp[2], swp[2];
    if constexpr (LW) {
        if (loader) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int par = (LW >= 2) ? (sw_id & 1) : pp;   // two / four loaders: instruction parity = loader id & 1 (both slots hold it)
                const unsigned sw = ((lane & 7) ^ ((4 * par + (lane >> 4)) & 7)) * 8;        // swizzled source chunk, in bf16 elements (x 2 = bytes)
                swp[pp] = sw;
                woffp[pp] = (unsigned)(n0l + par * 8 + lrow) * (unsigned)p.ldw * (unsigned)EB + sw * 2u;
                wmaxp[pp] = (unsigned)(p.N - 1) * (unsigned)p.ldw * (unsigned)EB + sw * 2u;
                aoffp[pp] = (unsigned)(m0l + par * 8 + lrow) * (unsigned)p.lda * (unsigned)EB + sw * 2u;
                amaxp[pp] = (unsigned)(p.M - 1) * (unsigned)p.lda * (unsigned)EB + sw * 2u;
            }
            if constexpr (CONV) {
#pragma unroll
                for (int r = 0; r < RA; ++r) {
                    int m = m0l + slot_a(r) * 8 + lrow; if (m > p.M - 1) m = p.M - 1;      // (the rows of THIS loader's r-th instruction)
                    conv_row(r, m, swp[r & 1] * 2u);
                }
            }
        }
    } else
    if (stager && !PHL) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int idx = slot_w(r);
        const int sw = ((lane & 7) ^ ((4 * idx + (lane >> 4)) & 7)) * 8;    // swizzled source chunk (elements)
        int n = n0l + idx * 8 + lrow; if (n > p.N - 1) n = p.N - 1;
        woff[r] = (unsigned)n * (unsigned)p.ldw * (unsigned)EB + (unsigned)sw * 2u;
    }
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int idx = slot_a(r);
        const int sw = ((lane & 7) ^ ((4 * idx + (lane >> 4)) & 7)) * 8;
        int m = m0l + idx * 8 + lrow; if (m > p.M - 1) m = p.M - 1;
        asw[r] = sw;
        if constexpr (CONV) {
            conv_row(r, m, (unsigned)sw * 2u);
            aoff[r] = 0;
        } else {
            aoff[r] = (unsigned)m * (unsigned)p.lda * (unsigned)EB + (unsigned)sw * 2u;
        }
    }
    }

    // ---- PH: a slice is (BM + BN) rows of 64 bytes; one LDS-DMA instruction covers 16 rows (lane -> row lane >> 2,
    // 16-byte position lane & 3), and position q of row r holds source chunk q ^ ((r >> 2) & 3): the 16-lane service
    // groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) then touch 16 distinct 16-byte bank groups.
    constexpr int PA = PHL ? BM / 16 / NW : 1, PB = PHL ? (BN / 16 + NW - 1) / NW : 1;      // DMA instructions per wave per slice
    auto ph_w = [&](int r) { int i = r * NW + w; if constexpr ((BN / 16) % NW != 0) i = min(i, BN / 16 - 1); return i; };     // W piece of this wave's r-th slot
    unsigned phA[PA], phW[PB];
    if constexpr (PHL) {
        const unsigned ch = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * (16 / EB));  // swizzled source chunk (elements)
#pragma unroll
        for (int r = 0; r < PA; ++r) {
            int m = m0l + (r * NW + w) * 16 + (lane >> 2); if (m > p.M - 1) m = p.M - 1;
            phA[r] = ((unsigned)m * (unsigned)p.lda + ch) * (unsigned)EB;
        }
#pragma unroll
        for (int r = 0; r < PB; ++r) {
            int n = n0l + ph_w(r) * 16 + (lane >> 2); if (n > p.N - 1) n = p.N - 1;
            phW[r] = ((unsigned)n * (unsigned)p.ldw + ch) * (unsigned)EB;
        }
    }
    auto ph_stage = [&](int slot, int s) {
        char* sA = smem + slot * SLOT;
        char* sW = sA + BM * 64;
#pragma unroll
        for (int r = 0; r < PA; ++r) blds16(rsA, phA[r], (unsigned)s * 64u, sA + (r * NW + w) * 1024);
#pragma unroll
        for (int r = 0; r < PB; ++r) blds16(rsW, phW[r], (unsigned)s * 64u, sW + ph_w(r) * 1024);
    };

    const int nk = p.K / (F8L ? 2 * BK : BK);          // (fp8: a staged row of 128 bytes holds 128 K values)
    int tap = 0, cc = 0;                              // conv K-tile cursor: tap (0..8), 64-channel chunk
    int cpt = CONV ? p.Cin / (F8L ? 2 * BK : BK) : 1;                  // (SC: the chunks of the CURRENT tap -- the shortcut taps 9 / 10 have their own channel counts)
    const int ntaps_all = SC ? p.ntaps + (p.c1s > 0) + (p.c2s > 0) : p.ntaps;

    // conv: the per-lane byte offset of a tap is computed once per tap (when the 64-channel cursor cc wraps);
    // the channel chunk rides in the scalar soffset.  Padding taps get an offset beyond num_records (-> zeros).
    unsigned cvo[RA];
    constexpr int SRW = (CONV && SCP) ? BM / SW : 1;   // fp8 conv: scale rows this staging wave gathers (row sw_id * SRW + lane)
    int sb_ = 0, sy_ = 0, sx_ = 0; unsigned scv = 0x80000000u;
    if constexpr (CONV && SCP) {
        if (stager) {
            const int hw = p.Ho * p.Wo;
            int m = m0l + sw_id * SRW + min(lane, SRW - 1); if (m > p.M - 1) m = p.M - 1;
            sb_ = m / hw; const int rem = m - sb_ * hw;
            sy_ = rem / p.Wo; sx_ = rem - sy_ * p.Wo;
        }
    }
    auto conv_tap_offsets = [&]() __attribute__((always_inline)) {      // (outlined, it takes the per-row arrays through scratch memory)
        if constexpr (SC) {
            if (tap >= p.ntaps) {                      // shortcut tap: the output pixel itself, Cs channels (stride-1 geometry: H x W = Ho x Wo)
                const int Cs = tap == p.ntaps ? p.c1s : p.c2s;
#pragma unroll
                for (int r = 0; r < RA; ++r)
                    cvo[r] = (unsigned)(cpix[r] * Cs + (LW ? (int)swp[r & 1] : asw[LW ? 0 : r])) * 2u;
                cpt = Cs / BK;
                return;
            }
        }
        // TMIX_CONV_T3: a (3,1,1) kernel over the first (frame) axis only -- 3 taps, kx fixed at the centre
        const int ky = p.mode == TMIX_CONV_T3 ? tap : tap / 3, kx = p.mode == TMIX_CONV_T3 ? 1 : tap - ky * 3;
        const unsigned pixb = (unsigned)p.Cin * (unsigned)EB, rowb = (unsigned)p.Wd * pixb;         // bytes of a source pixel / of a source row
        if (p.mode == TMIX_CONV_UP2) {
            // nearest x2: source row = (uy0 + ky) >> 1 = (uy0 >> 1) + {0, parity of uy0, 1}[ky], columns alike
            const unsigned sd = (ky == 2 ? rowb : 0u) + (kx == 2 ? pixb : 0u), ay = ky == 1 ? rowb : 0u, ax = kx == 1 ? pixb : 0u;
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                const unsigned o = cbase[r] + sd + ((cmask[r] >> 16) & 1u) * ay + ((cmask[r] >> 17) & 1u) * ax;
                cvo[r] = ((cmask[r] >> tap) & 1u) ? o : 0x80000000u;
            }
        } else {
            const unsigned sd = (unsigned)ky * rowb + (unsigned)kx * pixb;
#pragma unroll
            for (int r = 0; r < RA; ++r) cvo[r] = ((cmask[r] >> tap) & 1u) ? cbase[r] + sd : 0x80000000u;
        }
        if constexpr (CONV && SCP) {                   // the same tap for this wave's scale rows: byte offset of the row's Cin / 32 scales
            int iy, ix; bool ok;
            if (p.mode == TMIX_CONV_S1 || p.mode == TMIX_CONV_T3) { iy = sy_ + ky - 1;     ix = sx_ + kx - 1;     ok = (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd); }
            else if (p.mode == TMIX_CONV_S2) { iy = 2 * sy_ + ky - 1; ix = 2 * sx_ + kx - 1; ok = (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd); }
            else if (p.mode == TMIX_CONV_S2A) { iy = 2 * sy_ + ky; ix = 2 * sx_ + kx; ok = (iy < p.H) & (ix < p.Wd); }
            else { const int uy = sy_ + ky - 1, ux = sx_ + kx - 1;
                   ok = (uy >= 0) & (uy < 2 * p.H) & (ux >= 0) & (ux < 2 * p.Wd); iy = uy >> 1; ix = ux >> 1; }
            scv = ok ? (unsigned)((sb_ * p.H + iy) * p.Wd + ix) * (unsigned)(p.Cin / 32) : 0x80000000u;
        }
    };
    if constexpr (CONV) { if (stager) conv_tap_offsets(); }

    // fp8 lock-step loop with MX-block A scales: the K-tile's 4 blocks x BM rows of E8M0 bytes as one 1 KB piece behind the W tile -- lane l carries bytes
    // [16 l, 16 l + 16) of it (block 16 l / BM, rows 16 l % BM ..), and staging wave s issues the lanes l % SW == s: one exec-masked instruction per wave,
    // so every staging wave issues the same L instructions per stage (counted vmcnt)
    const bool sc_mine = SCP && (CONV ? lane < SRW : (lane % SW == sw_id) && (16 * lane < 4 * BM));
    unsigned sc_voff = 0;
    __amdgpu_buffer_rsrc_t rsSc = rsA;
    if constexpr (SCP) {
        if constexpr (CONV) rsSc = __builtin_amdgcn_make_buffer_rsrc((void*)p.scaleA, 0, (int)(p.bytesA / 32), 0x00020000);      // [pixels][Cin / 32]
        else {
            rsSc = __builtin_amdgcn_make_buffer_rsrc((void*)p.scaleA, 0, (int)((int64_t)(p.K / 32) * p.ldScaleA), 0x00020000);
            sc_voff = (unsigned)((16 * lane) / BM) * (unsigned)p.ldScaleA + (unsigned)((16 * lane) % BM) + (unsigned)(bz * p.strideScaleA + m0l);
        }
    }
    auto scale_piece = [&](char* sA, int kt) __attribute__((always_inline)) {
        if constexpr (SCP && CONV) {
            // (kt unused: the cursor (tap, cc) is the conv's own; 4 bytes per row = the K-tile's four MX blocks, rows [sw_id * SRW, + SRW) of the tile)
            if (sc_mine) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsSc, (__attribute__((address_space(3))) void*)(sA + A_TILE + B_TILE + sw_id * SRW * 4), 4, scv, (unsigned)cc * 4u, 0, 0);
        } else if constexpr (SCP) { if (sc_mine) blds16(rsSc, sc_voff, (unsigned)kt * 4u * (unsigned)p.ldScaleA, sA + A_TILE + B_TILE); }
    };
    // piece r of this wave's share of K-tile kt (plain GEMM without loader waves): A pieces, W pieces, then the scale piece
    auto dma_piece = [&](const int r, char* sA, int kt) __attribute__((always_inline)) {
        if constexpr (!CONV && !LW) {
            if (r < RA) blds16(rsA, aoff[r], (unsigned)kt * (BK * 2), sA + slot_a(r) * 1024);
            else if (r < RA + RB) blds16(rsW, woff[r - RA], (unsigned)kt * (BK * 2), sA + A_TILE + slot_w(r - RA) * 1024);
            else scale_piece(sA, kt);
        }
    };
    auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
        char* sA = smem + buf * STAGE;
        char* sW = sA + A_TILE;
        bool a_done = false;
        // byte offset of this K-tile inside a weight row: the plain GEMM walks K in order; the convolution's cursor (tap, cc) walks CHANNEL-CHUNK major
        // (see the cursor advance below), weight rows are [tap][Cin] (+ the shortcut tensors' channels behind the taps)
        unsigned wk = (unsigned)kt * (BK * 2);
#ifndef TMIX_ABL_WSEQ     // (dev A/B builds: keep the sequential walk of the weight rows under the chunk-major A gather -- wrong results, timing valid)
        if constexpr (CONV && !LW) {
            const int tap_u = __builtin_amdgcn_readfirstlane(tap), cc_u = __builtin_amdgcn_readfirstlane(cc);
            if (SC && tap_u >= p.ntaps) wk = (unsigned)(p.ntaps * p.Cin + (tap_u > p.ntaps ? p.c1s : 0)) * 2u + (unsigned)cc_u * (BK * 2);
            else wk = (unsigned)(tap_u * p.Cin) * (unsigned)EB + (unsigned)cc_u * (BK * 2);
        }
#endif
        if constexpr (CONV && SC) {
            // the A source of this K-tile: the conv input for the nine taps, then the shortcut tensors.  The cursor is wave-uniform (said explicitly),
            // and the choice is made on plain pointers -- a select between buffer RESOURCES goes through scratch memory and waterfall loops
            const int tap_u = __builtin_amdgcn_readfirstlane(tap);
            const void* base = tap_u < p.ntaps ? (const void*)Ab : (tap_u == p.ntaps ? (const void*)p.S1 : (const void*)p.S2);
            const unsigned bytes = tap_u < p.ntaps ? p.bytesA : (tap_u == p.ntaps ? p.bytesS1 : p.bytesS2);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
            a_done = true;
#pragma unroll
            for (int r = 0; r < RA; ++r) blds16(rs, cvo[r], (unsigned)cc * (BK * 2), sA + slot_a(r) * 1024);
        }
        if (!a_done)
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            if constexpr (CONV) blds16(rsA, cvo[r], (unsigned)cc * (BK * 2), sA + slot_a(r) * 1024);
            else if constexpr (LW) blds16(rsA, min(aoffp[(r * LW) & 1] + (unsigned)(slot_a(r) >> 1) * (unsigned)(16 * EB * p.lda), amaxp[(r * LW) & 1]),
                                          (unsigned)kt * (BK * 2), sA + slot_a(r) * 1024);
            else                blds16(rsA, aoff[r], (unsigned)kt * (BK * 2), sA + slot_a(r) * 1024);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if constexpr (LW) blds16(rsW, min(woffp[(r * LW) & 1] + (unsigned)(slot_w(r) >> 1) * (unsigned)(16 * EB * p.ldw), wmaxp[(r * LW) & 1]),
                                     wk, sW + slot_w(r) * 1024);
            else              blds16(rsW, woff[r], wk, sW + slot_w(r) * 1024);
        }
        scale_piece(sA, kt);
        if constexpr (CONV) {
            // K order of the convolution (round 6): CHANNEL-CHUNK major, the taps of a 64-channel chunk back to back.  Tap-major (rounds 1-5) re-read an input pixel
            // Cin / 64 K-tiles after its neighbour tap had fetched it -- by then the XCD's 4 MB L2 had turned over (32 resident tiles x 36-72 KB per K-tile) and the
            // re-read went through the fabric: FETCH_SIZE 4.5 x algorithmic.  Now the kx neighbours are consecutive K-tiles and the ky neighbours three apart.  The
            // per-lane tap offsets are recomputed every K-tile (a dozen VALU instructions per staged row, under the MFMAs); the shortcut tensors keep their order.
            // (The loader-wave instantiations -- tiling 20 -- keep the tap-major walk: their two loaders are the launch's critical path, and with the offsets
            // derived every K-tile and the weight rows walked tap-strided they lost 42 % hot, 152 vs 107 us at 32 x 32 1280 -> 1280; tools/jobs6_wseq.sh.)
            if (LW || (SC && tap >= p.ntaps)) { if (++cc == cpt) { cc = 0; ++tap; if (tap < ntaps_all) conv_tap_offsets(); } }
            else {
                if (++tap == p.ntaps) { tap = 0; if (++cc == cpt) { cc = 0; tap = p.ntaps; } }
                if (tap < ntaps_all) conv_tap_offsets();
            }
        }
    };

    // ---- loader wave (LW): leaves here, before any math-wave state (accumulators, fragments) becomes live
    if constexpr (LW) {
        if (loader) {                                  // ---- loader wave: DMA issue + counted waits only
            // tile 0 is handed over as soon as it has landed with ONE more tile requested behind it; the remaining prologue tiles are
            // requested while the math waves already multiply (each LDS-DMA instruction costs this wave ~100 issue cycles: with all
            // NS - 1 tiles in front of the first barrier the math waves waited for 9-18 instructions they did not need yet)
            constexpr int PRE = NS >= 4 ? 2 : NS - 1;
#pragma unroll
            for (int s = 0; s < PRE; ++s)
                if (s < nk) stage(s, s);
            if (nk >= PRE) wait_vmcnt<(PRE - 1) * L>(); else wait_vmcnt<0>();
#pragma unroll
            for (int u = 0; u < PFU; ++u) asm volatile("" :: "v"(pf_keep[u]));
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int s = PRE; s < NS - 1; ++s)
                if (s < nk) stage(s, s);
            int nxt = NS - 1;
            for (int kt = 0; kt < nk; ++kt) {
                const bool more = kt + NS - 1 < nk;
                if (more) stage(nxt, kt + NS - 1);    // its ring slot was released by the barrier that ended iteration kt-1
                if (more) wait_vmcnt<(NS - 2) * L>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
            }
            if (p.stats_out) __syncthreads();          // the math waves' statistics reduction has one more barrier
            return;
        }
    }

    // ---- fused LayerNorm (consumer side), part 1: the producer GEMM left ln_parts partial {sum, sum of squares}
    // per row (one per column tile of ITS grid).  Thread t owns tile row t and tile column t: the partials (added in
    // a fixed order -> scheduling-independent) and the weight column sum are requested here, and reduced once the
    // prologue's K-tiles are in flight (ln_reduce), so their latency rides under the prologue.  The results go to a
    // small LDS block behind the staging ring, already in MFMA-operand form (see part 2).
    uint4* ln_mfrag = (uint4*)(smem + RING);          // [BM] -mean pieces, [BN] colsum pieces, then float rstd[BM]
    uint4* ln_cfrag = ln_mfrag + BM;
    float* ln_rs = (float*)(ln_cfrag + BN);
    // the tile's BN bias values (zeros without a bias) for the straight-line staged epilogue: fetched here, parked in LDS by ln_reduce --
    // a load issued between the epilogue's stores would queue behind them (vmcnt counts stores too), and 8 values per chunk kept in
    // registers across the last K-tiles pushed the 128 x 160 tilings past 256 VGPRs
    float* bias_lds = ln_rs + BM;
    float bias_r = 0.f;
    if (p.bias && tid < BN && n0 + tid < p.N) bias_r = (p.bias + (int64_t)bzw * p.strideBias)[n0 + tid];
    // convolution: the time-embedding row of the tile's image likewise (every tile of this path lies inside one image; a tile that
    // straddles two row groups takes the generic epilogue, which looks the row up per output row)
    float* rgb_lds = bias_lds + BN;
    float rgb_r = 0.f;
    bool rgb_one = true;
    if constexpr (CONV) {
        if (p.rgb) {
            const int g0 = m0 / p.rows_per_group;
            rgb_one = (min(m0 + BM, p.M) - 1) / p.rows_per_group == g0;
            if (rgb_one && tid < BN && n0 + tid < p.N) rgb_r = p.rgb[(int64_t)g0 * p.N + n0 + tid];
        }
    }
    constexpr int PU = 16;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    u32x2 lnv[PU];
    float ln_cs = 0.f;
    const bool ln_on = p.ln_stats != nullptr;
    if (ln_on) {
        // buffer loads: the per-lane offset is the row, the partial index rides in the scalar offset
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_stats + (int64_t)bz * p.strideLnStats), 0,
                                                                              (int)(p.ln_parts * p.ldLnStats * 8), 0x00020000);
        if (tid < BM) {
            const int lnm = min(m0 + tid, p.M - 1);
#pragma unroll
            for (int q = 0; q < PU; ++q)
                if (q < p.ln_parts) lnv[q] = __builtin_amdgcn_raw_buffer_load_b64(rsS, lnm * 8, q * (int)p.ldLnStats * 8, 0);
        }
        if (tid < BN) ln_cs = (p.ln_colsum + (int64_t)bzw * p.strideLnColsum)[min(n0 + tid, p.N - 1)];
    }
    // x = x1 + x2 + x3 (bf16 pieces by truncation, exact residuals): operand halves {x1,x1,x2,0 | x1,x3,x2,0} for -mean
    // and {x1,x2,x1,0 | x3,x1,x2,0} for colsum pair up to the six products x_a * y_b with a + b <= 4 (~24 bits)
    auto ln_reduce = [&]() {
        if (tid < BN) { bias_lds[tid] = bias_r; if constexpr (CONV) rgb_lds[tid] = rgb_r; }
        if (!ln_on) return;
        if (tid < BM) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int q = 0; q < PU; ++q)
                if (q < p.ln_parts) { s1 += __uint_as_float(lnv[q].x); s2 += __uint_as_float(lnv[q].y); }
            const float mean = s1 * p.ln_inv_c;
            ln_rs[tid] = rsqrtf(fmaxf(s2 * p.ln_inv_c - mean * mean, 0.f) + p.ln_eps);
            const float x = -mean;
            const unsigned x1 = __float_as_uint(x) & 0xffff0000u;
            const float r1 = x - __uint_as_float(x1);
            const unsigned x2 = __float_as_uint(r1) & 0xffff0000u;
            const unsigned x3 = __float_as_uint(r1 - __uint_as_float(x2)) & 0xffff0000u;
            ln_mfrag[tid] = make_uint4((x1 >> 16) | x1, x2 >> 16, (x1 >> 16) | x3, x2 >> 16);
        }
        if (tid < BN) {
            const unsigned x1 = __float_as_uint(ln_cs) & 0xffff0000u;
            const float r1 = ln_cs - __uint_as_float(x1);
            const unsigned x2 = __float_as_uint(r1) & 0xffff0000u;
            const unsigned x3 = __float_as_uint(r1 - __uint_as_float(x2)) & 0xffff0000u;
            ln_cfrag[tid] = make_uint4((x1 >> 16) | x2, x1 >> 16, (x3 >> 16) | x1, x2 >> 16);
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool trans = (EK == 1 || EK == 2 || EK == 4) ? false : (p.n_trans_begin >= 0) && (n0 >= p.n_trans_begin);
    // transposed tiles: square wave tiles swap the two LDS sources in the main loop (the lane then owns 4 consecutive m of one
    // n); every other tiling accumulates as usual and transposes while staging the stores through LDS
    constexpr bool SQ = (FM == FN && TM == TN) && !PH;            // (fp8 in either loop: no operand swap)
    const bool tswap = trans && SQ;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int fsw = (lane >> 1) & 7;                  // f(row) for fragment rows base + (lane & 31)

    // One MFMA body for both store orientations: acc[i][j] = mfma(b[j], a[i]) yields D[rows of b][rows of a], and
    // a lane holds 4 consecutive D-rows for one D-column.  Row-major tiles take a = A-tile, b = W-tile fragments
    // (lane: 4 consecutive n for one m); transposed tiles swap the two LDS sources (square wave tiles only), so
    // acc[i][j] then belongs to W-fragment i x A-fragment j and a lane holds 4 consecutive m for one n.
    const int offA = (wr * TM + l31) * 128, offW = A_TILE + (wc * TN + l31) * 128;
    const int off_a = tswap ? offW : offA, off_b = tswap ? offA : offW;
    // residual rows are requested just before the last K-tile's MFMAs so their latency hides behind compute
    const bf16_t* Rb = p.R ? p.R + (int64_t)bz * p.strideR : nullptr;
    const bool plain_epi = EK ? false : !trans && p.epilogue != TMIX_EPI_GEGLU && !(p.wide & 1);      // (the narrow plain form)
    constexpr bool PREF = FM * FN <= 4;               // large wave tiles have no registers to spare for it
    uint2 rres[PREF ? FM : 1][PREF ? FN : 1][4];
    auto prefetch_residual = [&]() {
        if constexpr (PREF)
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int m = m0 + wr * TM + i * 32 + l31; if (m > p.M - 1) m = p.M - 1;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int n = n0 + wc * TN + j * 32 + g * 8 + lhi * 4; if (n > p.N - 4) n = p.N - 4;
                    rres[i][j][g] = *(const uint2*)(Rb + (int64_t)m * p.ldr + n);
                }
        }
    };
    // the same for the LDS-staged (wide) plain epilogue, in ITS lane order -- 16 bytes per lane, chunk by chunk as `chunk` below walks
    // the tile.  Without it every chunk of a tile waits a full memory round trip for its residual rows before it can store (the
    // residual and C may be the same tensor, so the compiler cannot hoist those loads over the previous chunk's stores): timeline
    // of 4096 x 1280 x 1280 + residual, 128 x 160 tiles: epilogue 8.1 us of a 27.6 us workgroup.  Wave tiles up to 12 pieces.
    constexpr int WP_I = (FN / 2) * 4 + (FN & 1) * 2;          // 16-byte residual pieces per lane per 32-row block
    constexpr bool WPREF = FM * WP_I <= 12 && !(CONV && F8L && LW);      // (the loader-wave fp8 convolution has no registers for it: 16 dwords of scratch otherwise)
    uint4 rw[WPREF ? FM * WP_I : 1];
    const bool wide_res = EK != 1 && WPREF && Rb && (EK >= 2 || (p.wide & 1)) && kg == 0 && !(ABL & 64);
    auto prefetch_residual_wide = [&]() {
        if constexpr (WPREF) {
            auto grab = [&](int j0, int base, auto cf_tag) {
                constexpr int CF = decltype(cf_tag)::value, CW = CF * 32, LPR = CW / 8, RPI = 64 / LPR, NP = 32 / RPI;
                const int rr = lane / LPR, nc = n0 + wc * TN + j0 * 32 + (lane % LPR) * 8;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int ps = 0; ps < NP; ++ps) {
                        const int m = m0 + wr * TM + i * 32 + ps * RPI + rr;
                        if (m < p.M && nc < p.N) rw[i * WP_I + base + ps] = *(const uint4*)(Rb + (int64_t)m * p.ldr + nc);
                    }
            };
#pragma unroll
            for (int c = 0; c < FN / 2; ++c) grab(c * 2, c * 4, std::integral_constant<int, 2>{});
            if constexpr (FN & 1) grab(FN - 1, (FN / 2) * 4, std::integral_constant<int, 1>{});
        }
    };
    // in front of the last K-tile's MFMAs.  A launch without residual gets zeros there instead (the straight-line epilogue adds them
    // unconditionally); the registers become live here, not in front of the K loop.
    auto epi_prefetch = [&]() __attribute__((always_inline)) {
        if constexpr (WPREF) {
            if (wide_res) prefetch_residual_wide();
            else {
#pragma unroll
                for (int q = 0; q < FM * WP_I; ++q) rw[q] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    // ---- software pipeline: NS-1 tiles requested ahead, NS-2 stay in flight across each barrier
    if constexpr (PHL) {
        constexpr int PL = PA + PB;                    // this wave's DMA instructions per slice
        static_assert(2 * PL <= 63, "vmcnt immediate");
        const int ns = p.K / (64 / EB);
        const int offA4 = (wr * TM + l31) * 64, offW4 = BM * 64 + (wc * TN + l31) * 64;
        typedef int v8i_t __attribute__((ext_vector_type(8)));
        int f8sA[F8 ? FM : 1], f8sW[F8 ? FN : 1];      // fp8: this lane's row scales, the E8M0 byte replicated into all four byte lanes
        if constexpr (F8) {
            const unsigned char* sa = p.scaleA + (int64_t)bz * p.strideScaleA;
            const unsigned char* sw = p.scaleW + (int64_t)bzw * p.strideScaleW;
#pragma unroll
            for (int i = 0; i < FM; ++i) f8sA[i] = F8B ? 0 : (int)(sa[min(m0 + wr * TM + i * 32 + l31, p.M - 1)] * 0x01010101u);
#pragma unroll
            for (int j = 0; j < FN; ++j) f8sW[j] = (int)(sw[min(n0 + wc * TN + j * 32 + l31, p.N - 1)] * 0x01010101u);
        }
        // F8B: the tile's A scales -- bytes [k-block][row], K/32 <= SCB blocks -- are copied into LDS behind the fused-LayerNorm block by
        // 1 KB LDS-DMA pieces queued AHEAD of the prologue's operand pieces (so the prologue's counted wait retires them too); a lane
        // then reads byte [2 s + half][its row] in the LOAD segment of slice s.  (Streaming them through registers with
        // global_load_ubyte one slice ahead was tried first: the compiler copies loop-carried registers at the back edge, i.e.
        // while such a load is still in flight -- nothing it can see orders that copy behind the counted wait.)
        char* const scl = smem + RING + (BM + BN) * 16 + BM * 4 + BN * 4;          // behind the fused-LayerNorm block and the bias block
        if constexpr (F8B) {
            const int nblk = p.K / 32;
            const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)p.scaleA, 0, (int)((int64_t)nblk * p.ldScaleA), 0x00020000);
            const unsigned lane_off = (unsigned)((lane * 16) / BM) * (unsigned)p.ldScaleA + (unsigned)((lane * 16) % BM)
                                    + (unsigned)(bz * p.strideScaleA + m0);
            const int npcs = (nblk * BM + 1023) / 1024;
            for (int q = w; q < npcs; q += NW)
                blds16(rsS, lane_off + (unsigned)(q * (1024 / BM)) * (unsigned)p.ldScaleA, 0u, scl + q * 1024);
        }
        const int f4 = (lane >> 2) & 3;                // (row >> 2) & 3 of fragment row base + (lane & 31)
        const int q0 = ((0 + lhi) ^ f4) << 4, q1 = ((2 + lhi) ^ f4) << 4;      // k-step 0 / 1 of the slice
        const int grp = w >> 2;                        // waves 4-7 (the second wave of each SIMD) run one barrier behind
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < ns) ph_stage(s, s);
        ln_reduce();
        if (ns >= 3) wait_vmcnt<2 * PL>(); else if (ns == 2) wait_vmcnt<PL>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (prof_on) pt1 = prof_now();
#pragma unroll
        for (int u = 0; u < PFU; ++u) asm volatile("" :: "v"(pf_keep[u]));        // the prefetch touches have returned (older than the tiles waited for)
        if (grp) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        // One slice: LOAD segment (fragment reads of slice s; the wait that makes slice s + 1 visible), barrier, MFMA
        // segment, barrier.  The LDS-DMA instructions of slice s + 3 are issued INSIDE the MFMA cluster, one after every
        // fourth MFMA: among MFMAs a DMA costs ~60 issue cycles and about half of it hides behind the 32-cycle matrix
        // op, while beside the ds_reads of the LOAD segment each cost 100-185 cycles and made that segment (not the MFMA
        // segment of the partner wave) the length of every barrier interval.  Its ring slot is the one slice s - 1 was
        // read from, released two barriers ago.
        auto slice = [&](int s, auto stage_tag) {
            constexpr bool ST = decltype(stage_tag)::value;
            const char* pa = smem + (s & 3) * SLOT + offA4;
            const char* pw = smem + (s & 3) * SLOT + offW4;
            frag_ab a[2][FM], b[2][FN];
            v8i_t a8[F8 ? FM : 1], b8[F8 ? FN : 1];
            int f8use[F8B ? FM : 1];
            if constexpr (F8) {
                // operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (probed, tools/probe_mx.py): lane (row, half h) holds K values
                // [16 h, 16 h + 16) in its first 16 bytes and [32 + 16 h, 32 + 16 h + 16) in its second; the scale supplied by
                // lane half b covers K block [32 b, 32 b + 32) -- i.e. the first / second 16 bytes of BOTH halves
                const int c0 = (lhi ^ f4) << 4, c1 = ((2 + lhi) ^ f4) << 4;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const uint4 lo = *(const uint4*)(pa + i * 32 * 64 + c0), hi = *(const uint4*)(pa + i * 32 * 64 + c1);
                    a8[i] = (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const uint4 lo = *(const uint4*)(pw + j * 32 * 64 + c0), hi = *(const uint4*)(pw + j * 32 * 64 + c1);
                    b8[j] = (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                }
            } else {
#pragma unroll
            for (int i = 0; i < FM; ++i) { a[0][i] = *(const frag_ab*)(pa + i * 32 * 64 + q0); a[1][i] = *(const frag_ab*)(pa + i * 32 * 64 + q1); }
#pragma unroll
            for (int j = 0; j < FN; ++j) { b[0][j] = *(const frag_ab*)(pw + j * 32 * 64 + q0); b[1][j] = *(const frag_ab*)(pw + j * 32 * 64 + q1); }
            }
            // slice s + 1 must have landed (for every wave) before the barrier in front of its first reader; in flight
            // here: slices s + 1 and s + 2 (slice s + 3 is issued below)
            if constexpr (F8B) {
#pragma unroll
                for (int i = 0; i < FM; ++i)                 // the E8M0 byte goes to all four byte lanes of the scale operand
                    f8use[i] = (int)((unsigned)(unsigned char)scl[(2 * s + lhi) * BM + wr * TM + i * 32 + l31] * 0x01010101u);
            }
            if (s + 2 < ns) wait_vmcnt<PL>(); else wait_vmcnt<0>();
            if (PREF && s == ns - 1 && Rb && plain_epi) prefetch_residual();     // rides under the last MFMA segment
            if (s == ns - 1) epi_prefetch();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA segment
            __builtin_amdgcn_s_setprio(1);
            char* sA = smem + ((s + 3) & 3) * SLOT;
            char* sW = sA + BM * 64;
            constexpr int KSTEPS = F8 ? 1 : 2;                      // MFMA k-steps per slice
            constexpr int NMF = KSTEPS * FM * FN, GAP = NMF / PL;       // one DMA after every GAP-th MFMA
            int issued = 0;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        if constexpr (F8) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j], a8[i], acc[i][j], 0, 0, 0, f8sW[j], 0, F8B ? f8use[i] : f8sA[i]);
                        else if constexpr (ABL & 2) asm volatile("" :: "v"(b[kk][j]), "v"(a[kk][i]));
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0);
                        const int idx = (kk * FM + i) * FN + j;
                        if constexpr (ST) {
                            if (idx % GAP == (GAP > 1 ? 1 : 0) && issued < PL) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (issued < PA) blds16(rsA, phA[issued], (unsigned)(s + 3) * 64u, sA + (issued * NW + w) * 1024);
                                else             blds16(rsW, phW[issued - PA], (unsigned)(s + 3) * 64u, sW + ph_w(issued - PA) * 1024);
                                __builtin_amdgcn_sched_barrier(0);
                                ++issued;
                            }
                        }
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
        };
        int s = 0;
        if constexpr (!(ABL & 4)) for (; s + 3 < ns; ++s) slice(s, std::true_type{});
        for (; s < ns; ++s) slice(s, std::false_type{});
        if (!grp) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    } else
    {
    // ---- 4-wave / split-K / loader-wave tilings: software-pipelined K loop.
    // These tilings run ONE math wave per SIMD (two with KS = 2), so nothing hides a wave's own LDS latency.  Left to itself the
    // compiler placed every ds_read_b128 one or two instructions in front of the MFMA that consumes it (`s_waitcnt lgkmcnt(1)`
    // before nearly every MFMA, four fragment registers recycled) and the round-2 loop ran fragment reads, MFMAs and LDS-DMA issue
    // strictly one after the other -- ablations on 4096 x 1280 x 1280, 128 x 160 tiles (tools/jobs/r3b_ablate.sh): 18.8 us of K
    // loop, 14.3 without the DMA, 15.0 without the MFMAs, 8.2 with neither, against 5.3 us of MFMA time.  Here the fragments of
    // k-step kk + 1 are requested between the MFMAs of k-step kk (issue order pinned by sched_barrier), the K-tile hand-over
    // (all fragments in registers, counted vmcnt, barrier) sits in FRONT of the last k-step's MFMAs with the next tile's first
    // fragments requested right behind it, and a staging wave spreads the LDS-DMA instructions of tile kt + NS - 1 over the
    // MFMAs of k-step 0.  With loader waves (LW) the math waves run the same loop without any VMEM instruction.
    if constexpr (!LW) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nk) stage(s, s);
    }
    ln_reduce();
    if constexpr (!LW) { if (nk >= NS - 1) wait_vmcnt<(NS - 2) * L>(); else wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (prof_on) pt1 = prof_now();
#pragma unroll
    for (int u = 0; u < PFU; ++u) asm volatile("" :: "v"(pf_keep[u]));            // the prefetch touches have returned (older than the tiles waited for)
    int cur = 0, nxt = NS - 1;                        // ring positions of tile kt and tile kt+NS-1
    constexpr int KPW = F8L ? 2 : 4 / KS;              // k-steps of this wave's split-K group (fp8: two 64-wide k-steps per 128-byte row)
    const int k0 = kg * KPW;
    typedef int v8i_t __attribute__((ext_vector_type(8)));
    typedef std::conditional_t<F8L, v8i_t, frag_ab> frag_t;
    frag_t fa[2][FM], fb[2][FN];
    int f8a[2][(F8L && F8B) ? FM : 1];                 // MX-block scale of the A fragment (E8M0 byte in all four byte lanes), double-buffered with it
    int f8sA[(F8L && !F8B) ? FM : 1], f8sW[F8L ? FN : 1];
    if constexpr (F8L) {
        const unsigned char* swp_ = p.scaleW + (int64_t)bzw * p.strideScaleW;
#pragma unroll
        for (int j = 0; j < FN; ++j) f8sW[j] = (int)(swp_[min(n0 + wc * TN + j * 32 + l31, p.N - 1)] * 0x01010101u);
        if constexpr (!F8B) {
            const unsigned char* sap_ = p.scaleA + (int64_t)bz * p.strideScaleA;
#pragma unroll
            for (int i = 0; i < FM; ++i) f8sA[i] = (int)(sap_[min(m0 + wr * TM + i * 32 + l31, p.M - 1)] * 0x01010101u);
        }
    }
    // one operand fragment of k-step kk from a row base inside the ring (operand layout of the scaled MFMA: tools/probe_mx.py, see the phase-offset loop)
    auto rdfrag = [&](const char* row, int kk) __attribute__((always_inline)) -> frag_t {
        if constexpr (F8L) {
            const uint4 lo = *(const uint4*)(row + (((4 * kk + lhi) ^ fsw) << 4)), hi = *(const uint4*)(row + (((4 * kk + 2 + lhi) ^ fsw) << 4));
            return (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        } else {
            return *(const frag_ab*)(row + ((((k0 + kk) * 2 + lhi) ^ fsw) << 4));
        }
    };
    auto rdscale = [&](int rbuf, int kk, int i) __attribute__((always_inline)) -> int {
        if constexpr (CONV) return (int)((unsigned)(unsigned char)smem[rbuf * STAGE + A_TILE + B_TILE + (wr * TM + i * 32 + l31) * 4 + 2 * kk + lhi] * 0x01010101u);
        else return (int)((unsigned)(unsigned char)smem[rbuf * STAGE + A_TILE + B_TILE + (2 * kk + lhi) * BM + wr * TM + i * 32 + l31] * 0x01010101u);
    };
    auto mma = [&](f32x16& c, const frag_t& b_, const frag_t& a_, int j, int i, int S) __attribute__((always_inline)) {
        if constexpr (ABL & 2) asm volatile("" :: "v"(b_), "v"(a_));
        else if constexpr (F8L) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b_, a_, c, 0, 0, 0, f8sW[j], 0, F8B ? f8a[S][F8B ? i : 0] : f8sA[F8B ? 0 : i]);
        else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_, a_, c, 0, 0, 0);
    };
    constexpr int NMF = FM * FN, NRD = FM + FN;
    // One k-step with its issue order PINNED (a sched_barrier after every micro-group): MFMA q on register set S, then that
    // MFMA's share of the NEXT k-step's fragment reads (ring slot rbuf, k-step rkk, into set 1 - S: A fragments first, they feed
    // every MFMA of the row) and, when nd > 0, of the nd LDS-DMA instructions of K-tile dkt into ring slot dbuf.  S and nd are
    // constants at every call site (the loops are fully unrolled), so all register-array indices fold.
    // ROLL (tilings at the register limit: 256 x 320 over eight waves): MFMAs run column by column (j-major), the W fragments are
    // SINGLE-buffered -- fragment j is re-read for the next k-step right behind its last MFMA of this one, FM * (FN - 1 - j) + FM
    // MFMAs ahead of its next use -- and only the FM A fragments are double-buffered: 2 * FM + FN fragment registers instead of
    // 2 * (FM + FN), which is what keeps that tiling free of scratch spills (spill traffic would also break the counted vmcnt).
    constexpr bool ROLL = (NW * KS > 4 && NMF * 16 + 2 * NRD * 4 > 200) || (F8L && LW);
    auto kstep = [&](const int S, const int rbuf, const int rkk, const int nd, const int dbuf, const int dkt) {
        const char* pa = smem + rbuf * STAGE + off_a;
        const char* pb = smem + rbuf * STAGE + off_b;
        char* sA = smem + dbuf * STAGE;
        constexpr int RPQ = (NRD + NMF - 1) / NMF;
        const int dpq = (nd + NMF - 1) / NMF;
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
            const int i = ROLL ? q % FM : q / FN, j = ROLL ? q / FM : q - (q / FN) * FN;
            const int SB = ROLL ? 0 : S;
            mma(acc[i][j], fb[SB][j], fa[S][i], j, i, S);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ROLL) {
                if (q < FM) { fa[1 - S][q] = rdfrag(pa + q * 32 * 128, rkk); if constexpr (F8L && F8B) f8a[1 - S][q] = rdscale(rbuf, rkk, q); }
                if (i == FM - 1) fb[0][j] = rdfrag(pb + j * 32 * 128, rkk);
            } else {
#pragma unroll
                for (int u = 0; u < RPQ; ++u) {
                    const int r = q * RPQ + u;
                    if (r < FM) { fa[1 - S][r] = rdfrag(pa + r * 32 * 128, rkk); if constexpr (F8L && F8B) f8a[1 - S][r] = rdscale(rbuf, rkk, r); }
                    else if (r < NRD) fb[1 - S][r - FM] = rdfrag(pb + (r - FM) * 32 * 128, rkk);
                }
            }
            if constexpr (!CONV && !LW) {
#pragma unroll
                for (int u = 0; u < dpq; ++u) {
                    const int r = q * dpq + u;
                    if (r < nd) dma_piece(r, sA, dkt);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one K-tile; MORE (compile time): tile kt + NS - 1 exists and this wave stages its share of it here.  (The convolution's
    // gather recomputes per-lane offsets at tap boundaries behind a branch: its LDS-DMA instructions stay in front of the k-steps.)
    auto ktile = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value && !(ABL & 4) && !LW;
        if constexpr (CONV && MORE) { stage(nxt, kt + NS - 1); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (!decltype(more_tag)::value) {   // residual rows of the epilogue: requested in front of the LAST K-tile
            if (PREF && kt == nk - 1 && Rb && plain_epi) prefetch_residual();
            if (kt == nk - 1) epi_prefetch();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kk = 0; kk < KPW - 1; ++kk)
            kstep(kk & 1, cur, kk + 1, (kk == 0 && MORE && !CONV) ? L : 0, nxt, kt + NS - 1);
        if constexpr (KPW == 1 && !CONV && MORE) { stage(nxt, kt + NS - 1); __builtin_amdgcn_sched_barrier(0); }
        // every fragment of tile kt is in registers (its ring slot may be restaged after the barrier); tile kt + 1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!LW) { if constexpr (MORE) wait_vmcnt<(NS - 2) * L>(); else { if (kt + 1 < nk) wait_vmcnt<0>(); } }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        cur = (cur + 1 == NS) ? 0 : cur + 1;
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        // (behind the last K-tile these reads fetch a stale ring slot into registers nobody uses: cheaper than a branch here)
        kstep((KPW - 1) & 1, cur, 0, 0, 0, 0);
    };
    static_assert(KPW == 1 || ((KPW - 1) & 1) == 1, "the next tile's first fragments go to register set 0");
#pragma unroll
    for (int r = 0; r < NRD; ++r) {
        if (r < FM) { fa[0][r] = rdfrag(smem + off_a + r * 32 * 128, 0); if constexpr (F8L && F8B) f8a[0][r] = rdscale(0, 0, r); }
        else fb[0][r - FM] = rdfrag(smem + off_b + (r - FM) * 32 * 128, 0);
    }
    int kt = 0;
    for (; kt + NS - 1 < nk; ++kt) ktile(kt, std::true_type{});
    for (; kt < nk; ++kt) ktile(kt, std::false_type{});
    }

    if constexpr (KS > 1) {
        // split-K hand-over: group 1 parks its partial tile in the staging ring (every wave has passed the loop's last barrier,
        // the ring is free), lane-linear 16-byte pieces at the position its group-0 twin (same wave tile) reads them from
        float4* xch = (float4*)smem + (size_t)wm_ * (FM * FN * 4 * 64) + lane;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        xch[((i * FN + j) * 4 + g) * 64] = make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        }
        __syncthreads();
        if (kg == 1) {                                 // done; keep the workgroup's barrier count (as the loader waves do)
            __syncthreads();                           // group 0 has read the partials
            if (p.stats_out) __syncthreads();          // the epilogue's row-statistics exchange
            return;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = xch[((i * FN + j) * 4 + g) * 64];
                    acc[i][j][g * 4] += v.x; acc[i][j][g * 4 + 1] += v.y; acc[i][j][g * 4 + 2] += v.z; acc[i][j][g * 4 + 3] += v.w;
                }
        __syncthreads();                               // the patches of the staged epilogue reuse this memory
    }
    if (prof_on) pt2 = prof_now();
    if constexpr (ABL & 8) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" :: "v"(acc[i][j]));
        if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
        return;
    }
    // ---------------------------------------------------------------- fused LayerNorm (consumer side), part 2
    // A was the raw row x; with W' = W*gamma:  Linear(LN(x))[m][n] = rstd_m * (acc[m][n] - mean_m * colsum_n) + t_n
    // (t_n arrives as the bias).  The rank-1 term -mean_m * colsum_n is one more MFMA k-step per fragment, fed from
    // the operand pieces ln_reduce left in LDS (visible: every wave has passed >= 2 barriers since); rstd_m is
    // applied as the multiplier of the bias FMA in the epilogue.
    float rs_row[FM];                                 // row-major tiles: rstd of this lane's row per fragment
#pragma unroll
    for (int i = 0; i < FM; ++i) rs_row[i] = 1.f;
    if (p.ln_stats) {
        // fragments follow the LDS sources of the main loop: a[i] <- rows of off_a's tile, b[j] <- rows of off_b's
        const uint2* srcA = (const uint2*)(ln_mfrag + wr * TM + l31) + lhi;
        const uint2* srcW = (const uint2*)(ln_cfrag + wc * TN + l31) + lhi;
        const uint2* src_a = tswap ? srcW : srcA;
        const uint2* src_b = tswap ? srcA : srcW;
        frag_ab fa[FM], fb[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const uint2 h = src_a[i * 64];
            uint4 u = make_uint4(h.x, h.y, 0u, 0u);
            fa[i] = *(frag_ab*)&u;
            if (!tswap) rs_row[i] = ln_rs[wr * TM + i * 32 + l31];
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const uint2 h = src_b[j * 64];
            uint4 u = make_uint4(h.x, h.y, 0u, 0u);
            fb[j] = *(frag_ab*)&u;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }

    // 32x32 accumulator: lane holds column (lane&31), rows (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const float* bias = p.bias ? p.bias + (int64_t)bzw * p.strideBias : nullptr;
    // ---- LDS-staged ("wide") stores.  In the accumulator layout a lane owns 4 consecutive outputs of ONE row, so a store
    // instruction scatters 64 x 8 bytes over 32 rows: 32 partial-line requests per instruction, and the epilogue of a
    // 256 x 256 tile took 14-24 us (a quarter to a half of the whole workgroup; tools/gemm_lab `tl`).  The staging ring is
    // free by now: each wave copies a 32-row block of its tile to a private LDS patch (rows padded by 16 bytes), reads it
    // back row-major -- 8 consecutive outputs per lane, 4-8 lanes per row -- and does bias / LayerNorm scale / activation /
    // residual there, so residual loads and C stores are 16 bytes per lane and 64-128 contiguous bytes per row.  Values and
    // operation order per element are those of the narrow path (bit-identical C); only the row statistics add in a new order.
    constexpr int STG_MAX = 32 * (2 * 32 * 4 + 16);               // bytes of LDS patch per wave (largest chunk: 64 fp32 columns)
    if (trans) {
        if constexpr (!SQ) {
            // accumulators in the usual orientation (lane: 4 consecutive n of row m = lane & 31): every 32 x (FM * 32) block
            // [n][m] is written to the wave's LDS patch two bytes at a time, read back as rows of Ct and stored 16 bytes per lane
            // (launch() only routes a transposed region here when the 16-byte form is allowed, p.wide & 4)
            bf16_t* Ct = p.Ct + (int64_t)bz * p.strideCt;
            constexpr int SR = FM * 64 + 16, LPR = FM * 4, RPI = 64 / LPR, NP = 32 / RPI;
            static_assert(32 * SR <= STG_MAX && 64 % LPR == 0, "transposed staging patch");
            char* stg = smem + w * STG_MAX;
            const int rr = lane / LPR, cc = (lane % LPR) * 8;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int nb = n0 + wc * TN + j * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (bias && nb < p.N) { const float4 b4 = *(const float4*)(bias + nb + g * 8 + lhi * 4); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            *(bf16_t*)(stg + (g * 8 + lhi * 4 + r) * SR + (i * 32 + l31) * 2) = f2bf(fmaf(acc[i][j][g * 4 + r], rs_row[i], bv[r]));
                }
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    const int r = ps * RPI + rr, n = nb + r, m = m0 + wr * TM + cc;
                    const uint4 v = *(const uint4*)(stg + r * SR + cc * 2);
                    if (n >= p.N || m >= p.M) continue;
                    bf16_t* dst = Ct + (int64_t)(n - p.n_trans_begin) * p.ldct + m;
                    if (m + 8 <= p.M) *(uint4*)dst = v;
                    else {
                        const unsigned u[4] = {v.x, v.y, v.z, v.w};
                        for (int k = 0; k < 8 && m + k < p.M; ++k) dst[k] = (bf16_t)(u[k >> 1] >> ((k & 1) * 16));
                    }
                }
            }
            if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
            return;
        }
        if constexpr (SQ) {
            bf16_t* Ct = p.Ct + (int64_t)bz * p.strideCt;
            if (EK == 3 || (p.wide & 4)) {
                constexpr int CF = (FN % 2 == 0) ? 2 : 1, SR = CF * 64 + 16, LPR = CF * 4, RPI = 64 / LPR, NP = 32 / RPI;
                char* stg = smem + w * STG_MAX;
                const int rr = lane / LPR, cc = (lane % LPR) * 8;
#pragma unroll
                for (int i = 0; i < FM; ++i) {             // W fragment: 32 rows of Ct
                    const int nl = n0 + wc * TN + i * 32 + l31;
                    const float bv = (bias && nl < p.N) ? bias[nl] : 0.f;
#pragma unroll
                    for (int c = 0; c < FN / CF; ++c) {
#pragma unroll
                        for (int jj = 0; jj < CF; ++jj)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int j = c * CF + jj, ml = wr * TM + j * 32 + g * 8 + lhi * 4;
                                const float4 rs = p.ln_stats ? *(const float4*)(ln_rs + ml) : make_float4(1.f, 1.f, 1.f, 1.f);
                                uint2 v;
                                v.x = pack_bf2(fmaf(acc[i][j][g * 4 + 0], rs.x, bv), fmaf(acc[i][j][g * 4 + 1], rs.y, bv));
                                v.y = pack_bf2(fmaf(acc[i][j][g * 4 + 2], rs.z, bv), fmaf(acc[i][j][g * 4 + 3], rs.w, bv));
                                *(uint2*)(stg + l31 * SR + (jj * 32 + g * 8 + lhi * 4) * 2) = v;
                            }
#pragma unroll
                        for (int ps = 0; ps < NP; ++ps) {
                            const int r = ps * RPI + rr, n = n0 + wc * TN + i * 32 + r, m = m0 + wr * TM + c * CF * 32 + cc;
                            const uint4 v = *(const uint4*)(stg + r * SR + cc * 2);
                            if (n >= p.N || m >= p.M) continue;
                            bf16_t* dst = Ct + (int64_t)(n - p.n_trans_begin) * p.ldct + m;
                            if (m + 8 <= p.M) *(uint4*)dst = v;
                            else {
                                const unsigned u[4] = {v.x, v.y, v.z, v.w};
                                for (int k = 0; k < 8 && m + k < p.M; ++k) dst[k] = (bf16_t)(u[k >> 1] >> ((k & 1) * 16));
                            }
                        }
                    }
                }
                if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
                return;
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {                 // W fragment (output row of Ct)
                const int n = n0 + wc * TN + i * 32 + l31;
                if (n >= p.N) continue;
                const float bv = bias ? bias[n] : 0.f;
                bf16_t* row = Ct + (int64_t)(n - p.n_trans_begin) * p.ldct;
#pragma unroll
                for (int j = 0; j < FN; ++j)               // A fragment (4 consecutive m per register group)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ml = wr * TM + j * 32 + g * 8 + lhi * 4, m = m0 + ml;
                        if (m >= p.M) continue;
                        const float4 rs = p.ln_stats ? *(const float4*)(ln_rs + ml) : make_float4(1.f, 1.f, 1.f, 1.f);
                        const float o0 = fmaf(acc[i][j][g * 4 + 0], rs.x, bv), o1 = fmaf(acc[i][j][g * 4 + 1], rs.y, bv);
                        const float o2 = fmaf(acc[i][j][g * 4 + 2], rs.z, bv), o3 = fmaf(acc[i][j][g * 4 + 3], rs.w, bv);
                        if (m + 3 < p.M && ((p.ldct & 3) == 0)) {
                            uint2 v;
                            v.x = pack_bf2(o0, o1);
                            v.y = pack_bf2(o2, o3);
                            *(uint2*)(row + m) = v;
                        } else {
                            const float o[4] = {o0, o1, o2, o3};
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (m + r < p.M) row[m + r] = f2bf(o[r]);
                        }
                    }
            }
        }
        if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
        return;
    }

    bf16_t* Cb = p.C + (int64_t)bz * p.strideC;
    if (EK == 1 || (EK == 0 && p.epilogue == TMIX_EPI_GEGLU)) {
        // weight rows are interleaved in 16-row groups [value_j | gate_j]: within a 32-row fragment, accumulator
        // register groups g=0,1 (rows 0-15) are the value half and g=2,3 (rows 16-31) the gate half.
        if (EK == 1 || (p.wide & 2)) {
            char* stg = smem + w * STG_MAX;
            auto chunk = [&](int i, int j0, auto cf_tag) {       // CF fragments -> CF * 16 output columns of one 32-row block
                constexpr int CF = decltype(cf_tag)::value, OC = CF * 16, SR = OC * 2 + 16, LPR = OC / 8, RPI = 64 / LPR, NP = 32 / RPI;
                const int rr = lane / LPR, cc = (lane % LPR) * 8;
#pragma unroll
                for (int jj = 0; jj < CF; ++jj) {
                    const int j = j0 + jj;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        float o[4];
                        // the tile's bias sits in LDS (zeros without one; parked by ln_reduce): no global load between the chunks' stores,
                        // no branch in the arithmetic
                        const float4 ba = *(const float4*)(bias_lds + wc * TN + j * 32 + g * 8 + lhi * 4), bg = *(const float4*)(bias_lds + wc * TN + j * 32 + g * 8 + lhi * 4 + 16);
                        const float bav[4] = {ba.x, ba.y, ba.z, ba.w}, bgv[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = fmaf(acc[i][j][g * 4 + r], rs_row[i], bav[r]), gt = fmaf(acc[i][j][(g + 2) * 4 + r], rs_row[i], bgv[r]);
                            o[r] = a * gelu_erf_f(gt);
                        }
                        uint2 v; v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]);
                        *(uint2*)(stg + l31 * SR + (jj * 16 + g * 8 + lhi * 4) * 2) = v;
                    }
                }
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    const int r = ps * RPI + rr, m = m0 + wr * TM + i * 32 + r;
                    const uint4 v = *(const uint4*)(stg + r * SR + cc * 2);
                    const int nb = n0 + wc * TN + (j0 + cc / 16) * 32;              // weight row of the fragment this lane's columns come from
                    const int col = (n0 + wc * TN) / 2 + j0 * 16 + cc;              // output column of this lane's 8 values
                    if constexpr (CF >= 2) {
                        if (p.f8out) {
                            // e4m3 output with one E8M0 scale per 32 columns (the MX block of the GEMM that reads it): the block is the
                            // 4 adjacent lanes of this row
                            const unsigned u[4] = {v.x, v.y, v.z, v.w};
                            float f[8], am = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(u[k] << 16); f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
                                                          am = fmaxf(am, fmaxf(fabsf(f[2 * k]), fabsf(f[2 * k + 1]))); }
                            am = quad_max(am);
                            const int e = e8m0_for_amax(am);
                            const float inv = exp2_neg_int(e);
                            int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false); q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, q0, true);
                            int q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, q1, true);
                            if (m >= p.M || nb >= p.N) continue;
                            *(uint2*)((unsigned char*)p.C + (int64_t)bz * p.strideC + (int64_t)m * p.ldc + col) = make_uint2((unsigned)q0, (unsigned)q1);
                            if ((lane & 3) == 0) p.scale_out[(int64_t)(col >> 5) * p.ldScaleOut + (int64_t)bz * p.M + m] = (unsigned char)(e + 127);
                            continue;
                        }
                    }
                    if (m >= p.M || nb >= p.N) continue;
                    *(uint4*)(Cb + (int64_t)m * p.ldc + col) = v;
                }
            };
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                constexpr int C4 = FN / 4, R4 = FN % 4;
#pragma unroll
                for (int c = 0; c < C4; ++c) chunk(i, c * 4, std::integral_constant<int, 4>{});
                if constexpr (R4 >= 2) chunk(i, C4 * 4, std::integral_constant<int, 2>{});
                if constexpr (R4 & 1) chunk(i, FN - 1, std::integral_constant<int, 1>{});
            }
            if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
            return;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wr * TM + i * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int nb = n0 + wc * TN + j * 32;               // weight-row index of this fragment
                if (nb >= p.N) continue;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int nv = nb + g * 8 + lhi * 4;
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a = acc[i][j][g * 4 + r], gt = acc[i][j][(g + 2) * 4 + r];
                        if (bias) { a = fmaf(a, rs_row[i], bias[nv + r]); gt = fmaf(gt, rs_row[i], bias[nv + 16 + r]); }
                        else { a *= rs_row[i]; gt *= rs_row[i]; }
                        o[r] = a * gelu_erf_f(gt);
                    }
                    uint2 v; v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]);
                    *(uint2*)(Cb + (int64_t)m * p.ldc + nb / 2 + g * 8 + lhi * 4) = v;
                }
            }
        }
        if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
        return;
    }
    // LayerNorm producer side: {sum, sum of squares} of every row of this tile AS STORED (bf16-rounded), reduced over
    // the wave's fragments, the lane pair and the WG's wave columns in a fixed order, then written (not accumulated)
    // to stats_out[tile_n][m] -- one partial per column tile; the consumer adds the tiles_n partials.
    float2* sto = (!CONV && p.stats_out) ? (float2*)(p.stats_out + (int64_t)bz * p.strideStatsOut) : nullptr;      // (a convolution has no LayerNorm behind it)
    if (EK >= 2 || (p.wide & 1)) {
        char* stg = smem + w * STG_MAX;
        float2* redw = (float2*)(smem + NW * STG_MAX);           // [WN][BM] row-statistics exchange, behind the patches
        const bool f32out = p.epilogue == TMIX_EPI_F32OUT;
        // row statistics: partial {sum, sum of squares} per (32-row block, pass); the 64-column chunks (8 lanes per row, 4
        // passes of 8 rows) and a trailing 32-column chunk (4 lanes per row, 2 passes of 16 rows) map lanes to rows differently
        float sa1[FM][4], sa2[FM][4], sb1[FM][2], sb2[FM][2];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { sa1[i][q] = 0.f; sa2[i][q] = 0.f; }
            sb1[i][0] = sb1[i][1] = sb2[i][0] = sb2[i][1] = 0.f;
        }
        // FL bit 0: the STRAIGHT-LINE form for the common case (bf16 output, no activation; FL bit 1 also keeps the row statistics, bit 2
        // also leaves the e4m3 copy of the stored rows).  With one wave per SIMD the epilogue is bound by VALU issue and by the latencies nobody hides
        // (tools/jobs/r3za_epi_abl.sh: 5.3 of its 6.7 us on 128 x 160 tiles remain with bias loads, residual and stores all removed), and the
        // generic form below has a dozen wave-uniform branches per pass, each a scheduling barrier and a fetch bubble.  Here bias and
        // residual come from registers filled in front of the last K-tile (zeros when the launch has none: x * 1 + 0 and x + 0 leave
        // every value as it was), a lane outside the tile is handled by predicating its store, and the passes of a chunk interleave.
        // GroupNorm producer side (cs_out; flavour bit 8): column sums and sums of squares of the values AS STORED, per 32-row block -- the
        // granularity at which a wave owns whole columns in every tiling, so there is no exchange between waves, no barrier, and the buffer
        // layout [M / 32][2][N] does not depend on the tiling.  A lane of the staged epilogue holds 8 columns of one row per pass: it adds
        // its rows up in cv (sums in 0-7, squares in 8-15), then the 8 (16) lanes that share its columns are reduced as a reduce-SCATTER:
        // every permlane swap settles two values at once (one ends in each half), so 16 values cost 8 + 4 swaps and 4 (8) DPP adds, and
        // lane (bit 5 = statistic, bit 4 = column half) ends with four adjacent columns of one plane: one 16-byte store.
        auto cs_flush = [&](float (&cv)[16], int mb, int nc, bool ncok, auto cf_tag) __attribute__((always_inline)) {
            constexpr int CF = decltype(cf_tag)::value;
            float t[8], u[4];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = swap32_sum(cv[k], cv[8 + k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = swap16_sum(t[k], t[k + 4]);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] += dpp_row<0x128>(u[k]);
            if constexpr (CF == 1) {                                 // 4 lanes per row: the lanes of a column also differ in bit 2
#pragma unroll
                for (int k = 0; k < 4; ++k) u[k] += dpp_row<0x124>(u[k]);
            }
            if (!(lane & (CF == 1 ? 12 : 8)) && mb < p.M && ncok)
                *(float4*)(p.cs_out + ((int64_t)(mb >> 5) * 2 + (lane >> 5)) * p.N + nc + ((lane >> 4) & 1) * 4) = make_float4(u[0], u[1], u[2], u[3]);
        };
        auto cs_add = [&](float (&cv)[16], const uint4& v) __attribute__((always_inline)) {
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = __uint_as_float(u[k] << 16), hi = __uint_as_float(u[k] & 0xffff0000u);
                cv[2 * k] += lo; cv[2 * k + 1] += hi;
                cv[8 + 2 * k] = fmaf(lo, lo, cv[8 + 2 * k]); cv[9 + 2 * k] = fmaf(hi, hi, cv[9 + 2 * k]);
            }
        };
        auto chunk = [&](int j0, auto cf_tag, auto fl_tag) __attribute__((always_inline)) {     // CF fragments = CF * 32 fp32 columns of every 32-row block
            constexpr int CF = decltype(cf_tag)::value, CW = CF * 32, SR = CW * 4 + 16, LPR = CW / 8, RPI = 64 / LPR, NP = 32 / RPI;
            constexpr int FL = decltype(fl_tag)::value;              // (tilings without the residual registers: launches without residual only)
            const int rr = lane / LPR, cc = (lane % LPR) * 8;
            const int nc = n0 + wc * TN + j0 * 32 + cc;
            const bool ncok = nc < p.N;
            if constexpr ((FL & 1) != 0) {
                constexpr int CI = CF == 2 ? 0 : (FN / 2) * 4;          // residual pieces of this chunk inside rw (plus (j0 / 2) * 4 for pairs)
                const bool lnf = p.ln_stats != nullptr;
                const float4 bq0 = *(const float4*)(bias_lds + wc * TN + j0 * 32 + cc), bq1 = *(const float4*)(bias_lds + wc * TN + j0 * 32 + cc + 4);
                const float bq[8] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w};
                float tq[8];                                            // convolution: the image's time-embedding row (zeros without one)
                if constexpr (CONV) {
                    const float4 t0 = *(const float4*)(rgb_lds + wc * TN + j0 * 32 + cc), t1 = *(const float4*)(rgb_lds + wc * TN + j0 * 32 + cc + 4);
                    tq[0] = t0.x; tq[1] = t0.y; tq[2] = t0.z; tq[3] = t0.w; tq[4] = t1.x; tq[5] = t1.y; tq[6] = t1.z; tq[7] = t1.w;
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int mb = m0 + wr * TM + i * 32;
                    float cv[(FL & 8) ? 16 : 1];
                    if constexpr ((FL & 8) != 0) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) cv[k] = 0.f;
                    }
#pragma unroll
                    for (int jj = 0; jj < CF; ++jj)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *(float4*)(stg + l31 * SR + (jj * 32 + g * 8 + lhi * 4) * 4) =
                                make_float4(acc[i][j0 + jj][g * 4 + 0], acc[i][j0 + jj][g * 4 + 1], acc[i][j0 + jj][g * 4 + 2], acc[i][j0 + jj][g * 4 + 3]);
#pragma unroll
                    for (int ps = 0; ps < NP; ++ps) {
                        const int r = ps * RPI + rr, m = mb + r;
                        const float4 v0 = *(const float4*)(stg + r * SR + cc * 4), v1 = *(const float4*)(stg + r * SR + cc * 4 + 16);
                        const float rsl = ln_rs[wr * TM + i * 32 + r];       // (valid LDS also without the fused LayerNorm; discarded then)
                        const float rs = lnf ? rsl : 1.f;
                        float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = fmaf(o[k], rs, bq[k]);
                        if constexpr (CONV) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) o[k] += tq[k];
                        }
                        if constexpr (WPREF) {
                            const uint4 rq = rw[i * WP_I + CI + (CF == 2 ? (j0 / 2) * 4 : 0) + ps];
                            const unsigned ru[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) { o[2 * k] += __uint_as_float(ru[k] << 16); o[2 * k + 1] += __uint_as_float(ru[k] & 0xffff0000u); }
                        }
                        uint4 v;
                        v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]); v.z = pack_bf2(o[4], o[5]); v.w = pack_bf2(o[6], o[7]);
                        const bool ok = m < p.M && ncok;
                        if (ok) *(uint4*)(Cb + (int64_t)m * p.ldc + nc) = v;
                        if constexpr (F8C && (FL & 4) != 0) {           // the e4m3 + MX-block copy of the row AS STORED (as in the generic form)
                            const unsigned u8[4] = {v.x, v.y, v.z, v.w};
                            float f[8], am = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(u8[k] << 16); f[2 * k + 1] = __uint_as_float(u8[k] & 0xffff0000u);
                                                          am = fmaxf(am, fmaxf(fabsf(f[2 * k]), fabsf(f[2 * k + 1]))); }
                            am = quad_max(am);
                            const int e = e8m0_for_amax(am);
                            const float inv = exp2_neg_int(e);
                            int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false); q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, q0, true);
                            int q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, q1, true);
                            if (ok) {
                                *(uint2*)(p.f8copy + ((int64_t)bz * p.M + m) * p.ldF8copy + nc) = make_uint2((unsigned)q0, (unsigned)q1);
                                if ((lane & 3) == 0) p.scale_out[(int64_t)(nc >> 5) * p.ldScaleOut + (int64_t)bz * p.M + m] = (unsigned char)(e + 127);
                            }
                        }
                        if constexpr ((FL & 2) != 0) {                  // statistics of the values AS STORED (same order as the generic form)
                            const unsigned u[4] = {v.x, v.y, v.z, v.w};
                            float a1 = 0.f, a2 = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float lo = __uint_as_float(u[k] << 16), hi = __uint_as_float(u[k] & 0xffff0000u);
                                a1 += lo + hi; a2 = fmaf(lo, lo, a2); a2 = fmaf(hi, hi, a2);
                            }
                            if (!ok) { a1 = 0.f; a2 = 0.f; }
                            if constexpr (CF == 2) { sa1[i][ps] += a1; sa2[i][ps] += a2; } else { sb1[i][ps] += a1; sb2[i][ps] += a2; }
                        }
                        if constexpr ((FL & 8) != 0) cs_add(cv, v);     // (M % 32 == 0: a 32-row block is inside the matrix or not stored at all)
                    }
                    if constexpr ((FL & 8) != 0) cs_flush(cv, mb, nc, ncok, cf_tag);
                }
                return;
            }
            float bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bv[k] = 0.f;
            if (bias && ncok && !(ABL & 16)) {
                const float4 b0 = *(const float4*)(bias + nc), b1 = *(const float4*)(bias + nc + 4);
                bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int mb = m0 + wr * TM + i * 32;
                float cv[(FL & 8) ? 16 : 1];
                if constexpr ((FL & 8) != 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) cv[k] = 0.f;
                }
                uint4 rv[NP];
                if (Rb && !(ABL & 64)) {
#pragma unroll
                    for (int ps = 0; ps < NP; ++ps) {
                        const int m = mb + ps * RPI + rr;
                        if constexpr (WPREF) rv[ps] = rw[i * WP_I + (CF == 2 ? (j0 / 2) * 4 : (FN / 2) * 4) + ps];     // requested under the last K-tile
                        else if (m < p.M && ncok) rv[ps] = *(const uint4*)(Rb + (int64_t)m * p.ldr + nc);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < CF; ++jj)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *(float4*)(stg + l31 * SR + (jj * 32 + g * 8 + lhi * 4) * 4) =
                            make_float4(acc[i][j0 + jj][g * 4 + 0], acc[i][j0 + jj][g * 4 + 1], acc[i][j0 + jj][g * 4 + 2], acc[i][j0 + jj][g * 4 + 3]);
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    const int r = ps * RPI + rr, m = mb + r;
                    const float4 v0 = *(const float4*)(stg + r * SR + cc * 4), v1 = *(const float4*)(stg + r * SR + cc * 4 + 16);
                    if (m >= p.M || !ncok) continue;
                    float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    if (bias) { const float rs = p.ln_stats ? ln_rs[wr * TM + i * 32 + r] : 1.f;
#pragma unroll
                                for (int k = 0; k < 8; ++k) o[k] = fmaf(o[k], rs, bv[k]); }
                    else if (p.ln_stats) { const float rs = ln_rs[wr * TM + i * 32 + r];
#pragma unroll
                                for (int k = 0; k < 8; ++k) o[k] *= rs; }
                    if (p.rgb) {
                        const float* rg = p.rgb + (int64_t)(m / p.rows_per_group) * p.N + nc;
                        const float4 b0 = *(const float4*)rg, b1 = *(const float4*)(rg + 4);
                        o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w; o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
                    }
                    if (p.epilogue == TMIX_EPI_GELU) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = gelu_erf_f(o[k]);
                    } else if (p.epilogue == TMIX_EPI_QUICKGELU) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = o[k] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * o[k]));
                    }
                    if (Rb && !(ABL & 64)) {
                        const unsigned u[4] = {rv[ps].x, rv[ps].y, rv[ps].z, rv[ps].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) { o[2 * k] += bf2f((bf16_t)(u[k] & 0xffff)); o[2 * k + 1] += bf2f((bf16_t)(u[k] >> 16)); }
                    }
                    if (f32out) {
                        float* dst = (float*)p.C + (int64_t)bz * p.strideC + (int64_t)m * p.ldc + nc;
                        *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
                        *(float4*)(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
                    } else {
                        uint4 v;
                        v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]); v.z = pack_bf2(o[4], o[5]); v.w = pack_bf2(o[6], o[7]);
                        if constexpr (ABL & 32) asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                        else *(uint4*)(Cb + (int64_t)m * p.ldc + nc) = v;
                        if constexpr (F8C) if (p.f8copy) {
                            // the next GEMM's A operand: the row AS STORED, as e4m3 with one E8M0 scale per 32 columns (MX block = the 4
                            // adjacent lanes of this row; M % 32 == 0 and N % 32 == 0, so a block's lanes are all here)
                            const unsigned u8[4] = {v.x, v.y, v.z, v.w};
                            float f[8], am = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(u8[k] << 16); f[2 * k + 1] = __uint_as_float(u8[k] & 0xffff0000u);
                                                          am = fmaxf(am, fmaxf(fabsf(f[2 * k]), fabsf(f[2 * k + 1]))); }
                            am = quad_max(am);
                            const int e = e8m0_for_amax(am);
                            const float inv = exp2_neg_int(e);
                            int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false); q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, q0, true);
                            int q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, q1, true);
                            *(uint2*)(p.f8copy + ((int64_t)bz * p.M + m) * p.ldF8copy + nc) = make_uint2((unsigned)q0, (unsigned)q1);
                            if ((lane & 3) == 0) p.scale_out[(int64_t)(nc >> 5) * p.ldScaleOut + (int64_t)bz * p.M + m] = (unsigned char)(e + 127);
                        }
                        if (sto) {                                 // statistics of the values AS STORED
                            const unsigned u[4] = {v.x, v.y, v.z, v.w};
                            float a1 = 0.f, a2 = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float lo = __uint_as_float(u[k] << 16), hi = __uint_as_float(u[k] & 0xffff0000u);
                                a1 += lo + hi; a2 = fmaf(lo, lo, a2); a2 = fmaf(hi, hi, a2);
                            }
                            if constexpr (CF == 2) { sa1[i][ps] += a1; sa2[i][ps] += a2; } else { sb1[i][ps] += a1; sb2[i][ps] += a2; }
                        }
                        if constexpr ((FL & 8) != 0) cs_add(cv, v);
                    }
                }
                if constexpr ((FL & 8) != 0) cs_flush(cv, mb, nc, ncok, cf_tag);
            }
        };
        // chunks of two fragments (64 columns: 8 lanes x 16 bytes per row), a last single one when FN is odd
        constexpr int C2 = FN / 2;
        auto chunks = [&](auto fl_tag) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < C2; ++c) chunk(c * 2, std::integral_constant<int, 2>{}, fl_tag);
            if constexpr (FN & 1) chunk(FN - 1, std::integral_constant<int, 1>{}, fl_tag);
        };
        const bool fastp = (WPREF || !Rb) && !f32out && p.epilogue == TMIX_EPI_NONE && (!p.rgb || (CONV && rgb_one)) && !(ABL & 0xf0);      // (ablation bit 7: the generic form only)
        bool f8q = false;
        if constexpr (F8C) f8q = p.f8copy != nullptr;
        // flavour bits: 1 straight-line, 2 row statistics, 4 e4m3 copy, 8 column statistics
        // (8: column statistics for a GroupNorm behind this launch -- never together with row statistics or the e4m3 copy, gemm_conv.hip)
        if constexpr (EK == 4) { if constexpr (F8C) { if (!sto) chunks(std::integral_constant<int, 5>{}); else chunks(std::integral_constant<int, 7>{}); } }     // (the host sends only such launches here)
        else if constexpr (CS) { if (!fastp) chunks(std::integral_constant<int, 8>{}); else chunks(std::integral_constant<int, 9>{}); }
        else if (!fastp) chunks(std::integral_constant<int, 0>{});
        else if (!f8q) { if (!sto) chunks(std::integral_constant<int, 1>{}); else chunks(std::integral_constant<int, 3>{}); }
        else { if constexpr (F8C) { if (!sto) chunks(std::integral_constant<int, 5>{}); else chunks(std::integral_constant<int, 7>{}); } }
        if (sto) {
            // a row's partials sit in the lanes that stored its columns: reduce over those lanes (fixed butterfly order), the
            // group's first lane owns the row; the 32-column chunk's owners then add to the same slot (same wave: LDS in order)
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if constexpr (C2 > 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float a1 = sa1[i][q], a2 = sa2[i][q];
#pragma unroll
                        for (int off = 1; off < 8; off <<= 1) { a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); }
                        if ((lane & 7) == 0) redw[wc * BM + wr * TM + i * 32 + q * 8 + (lane >> 3)] = make_float2(a1, a2);
                    }
                }
                if constexpr (FN & 1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float b1 = sb1[i][q], b2 = sb2[i][q];
#pragma unroll
                        for (int off = 1; off < 4; off <<= 1) { b1 += __shfl_xor(b1, off); b2 += __shfl_xor(b2, off); }
                        if ((lane & 3) == 0) {
                            float2* dst = redw + wc * BM + wr * TM + i * 32 + q * 16 + (lane >> 2);
                            if constexpr (C2 > 0) { const float2 t = *dst; *dst = make_float2(t.x + b1, t.y + b2); }
                            else *dst = make_float2(b1, b2);
                        }
                    }
                }
            }
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                float2 t = redw[tid];
#pragma unroll
                for (int c = 1; c < WN; ++c) { const float2 u = redw[c * BM + tid]; t.x += u.x; t.y += u.y; }
                sto[(int64_t)tile_n * p.ldStatsOut + m0 + tid] = t;
            }
        }
        if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
        return;
    }
    float2* red = (float2*)smem;                                // [WN][BM]: the staging ring is free by now
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wr * TM + i * 32 + l31;
        float s1 = 0.f, s2 = 0.f;
        if (m < p.M) {
        const float* rg = p.rgb ? p.rgb + (int64_t)(m / p.rows_per_group) * p.N : nullptr;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wc * TN + j * 32 + g * 8 + lhi * 4;
                if (n >= p.N) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[i][j][g * 4 + r];
                if (bias) { const float4 b4 = *(const float4*)(bias + n);
                            o[0] = fmaf(o[0], rs_row[i], b4.x); o[1] = fmaf(o[1], rs_row[i], b4.y);
                            o[2] = fmaf(o[2], rs_row[i], b4.z); o[3] = fmaf(o[3], rs_row[i], b4.w); }
                else if (p.ln_stats) { o[0] *= rs_row[i]; o[1] *= rs_row[i]; o[2] *= rs_row[i]; o[3] *= rs_row[i]; }
                if (rg)   { const float4 b4 = *(const float4*)(rg + n);   o[0] += b4.x; o[1] += b4.y; o[2] += b4.z; o[3] += b4.w; }
                if (p.epilogue == TMIX_EPI_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = gelu_erf_f(o[r]);
                } else if (p.epilogue == TMIX_EPI_QUICKGELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = o[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * o[r]));
                }
                if (Rb) {
                    uint2 rv;
                    if constexpr (PREF) rv = rres[i][j][g];
                    else rv = *(const uint2*)(Rb + (int64_t)m * p.ldr + n);
                    o[0] += bf2f((bf16_t)(rv.x & 0xffff)); o[1] += bf2f((bf16_t)(rv.x >> 16));
                    o[2] += bf2f((bf16_t)(rv.y & 0xffff)); o[3] += bf2f((bf16_t)(rv.y >> 16));
                }
                if (p.epilogue == TMIX_EPI_F32OUT) {
                    *(float4*)((float*)p.C + (int64_t)bz * p.strideC + (int64_t)m * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    uint2 v; v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]);
                    *(uint2*)(Cb + (int64_t)m * p.ldc + n) = v;
                    if (sto) {
                        const float r0 = __uint_as_float(v.x << 16), r1 = __uint_as_float(v.x & 0xffff0000u);
                        const float r2 = __uint_as_float(v.y << 16), r3 = __uint_as_float(v.y & 0xffff0000u);
                        s1 += (r0 + r1) + (r2 + r3);
                        s2 = fmaf(r0, r0, s2); s2 = fmaf(r1, r1, s2); s2 = fmaf(r2, r2, s2); s2 = fmaf(r3, r3, s2);
                    }
                }
            }
        }
        if (sto) {                                     // wave-uniform; lanes l and l^32 hold the two halves of row m
            s1 = xor32_sum(s1); s2 = xor32_sum(s2);
            if (!lhi) red[wc * BM + wr * TM + i * 32 + l31] = make_float2(s1, s2);
        }
    }
    if (sto) {
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float2 t = red[tid];
#pragma unroll
            for (int c = 1; c < WN; ++c) { const float2 u = red[c * BM + tid]; t.x += u.x; t.y += u.y; }
            sto[(int64_t)tile_n * p.ldStatsOut + m0 + tid] = t;
        }
    }
    if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
}

// ------------------------------------------------------------------------------------------- launch
struct TileCfg { int bm, bn; };
// cfg ids (tmix.h TMIX_TILE_*): 1 = 128x128 (4 waves, 2 stages, 2 WG/CU), 2 = 256x128 (8 waves, 3 stages),
// 3 = 128x128 (4 waves, 4 stages, 1 WG/CU), 4 = 256x256 (8 waves, 2 stages)
// 5 = 256x128 (4 waves of 128x64, 3 stages, 1 WG/CU), 6 = 256x256 (4 waves of 128x128, 2 stages, 1 WG/CU):
// one wave per SIMD with a large register tile -- on this chip instructions of co-resident waves do not overlap on a
// SIMD, so MFMA utilisation is set by MFMAs per non-MFMA instruction, i.e. by the wave tile.
// 7 = 128x160 (4 waves of 32x160, 2 stages): N = 1280 / 640 split into 160-wide tiles gives exactly 256 / 512 tiles
// for this path's M = 4096 / 16384 GEMMs, i.e. whole rounds on 256 CUs instead of 1.25 / 2.5.
// 8..11 = tilings 7, 2, 1, 4 with one extra LOADER wave (wave specialisation, see gemm_conv_kernel); the 4-wave tilings
// with 128-wide wave tiles (5, 6) have no registers for a fifth wave on one of the SIMDs
// 12 = tiling 7 (128x160) with a 4-deep ring (one workgroup per CU, three K-tiles in flight: the in-sequence loop is bound by
// memory latency x bytes in flight, and 160-wide tiles divide N = 1280 / 640 exactly)
// 13 = 64x160 over FIVE waves (each 64x32), 4-deep ring: 2048 x 1280 -- the half-batch launches of the 32x32 level -- is
// exactly 256 tiles, one per CU, where 128x128 leaves 96 CUs idle (160 tiles) and 128x160 half of them
// 14 = 256x320 over eight waves (wave tile 64x160): the GEGLU up-projection 2048 x 10240 is exactly 256 tiles, where
// 256x256 runs 320 (a quarter-full second round); 128x320 over four waves was tried and lost to 128x160 everywhere
// 15 = 32x160 over five waves (each 32x32): 1024 x 1280 -- one batch row per chain, the CFG-pair calls -- is 256 tiles
// (a 5-deep ring for 13 measured the same as the 4-deep one)
// 16 = 256x256, 17 = 256x128 with the PHASE-OFFSET mainloop (PH: eight waves, K slices of 32 through a four-slot ring, the
// second wave of every SIMD one barrier behind the first; GEMM only, no transposed region)
// 18 = tiling 12 (128x160, 4-deep ring) with in-workgroup split-K over two wave groups (KS = 2): eight waves stage, GEMM only
// 19 / 20 / 21 = tiling 12 (128x160, 4-deep ring) with one / two / FOUR LOADER waves next to the four math waves (GEMM only); with four
// every SIMD hosts one math wave and one loader, and a K-tile's 36 LDS-DMA instructions are nine per loader
// 22 = 256x320 with the PHASE-OFFSET mainloop (eight waves of 64x160; bf16 GEMM only): the lock-step 256x320 loop (14) stops all eight waves at every
// K-tile hand-over -- 79 % of the MFMA rate with the LDS-DMA ablated -- where this one keeps one wave of every SIMD in its MFMA segment
// 23 = 128x160 over 2 x 2 math waves of 64x80 on v_mfma_f32_16x16x32_bf16 + four loader waves (its own kernel: gemm_w22.hip)
// 24 = 256x320 (tiling 14's tile and arithmetic) on PERSISTENT workgroups: one per CU walks its tiles, the next tile's first K-tile requested under the last one (gemm_ff1p.hip)
// 25 = tiling 23 with the fourth loader wave as an L2 PREFETCHER (touches the tile's operand lines eight K-tiles ahead of the ring)
// 26 = 3x3 stride-1 convolution with the input halo patch resident in LDS (4 x 32 pixel tiles, channel-chunk-major K loop; its own kernel: gemm_convh.hip)
constexpr int NUM_CFG = 26;

template <int BM, int BN, int WM, int WN, int NS, int CONV, int LW = 0, int PH = 0, int KS = 1, int CS = 0, int EK = 0, int SC = 0>
int launch_cfg(Params& p, int batch, hipStream_t st) {
    static_assert(KS == 1 || BM * BN * 4 <= NS * (BM + BN) * 128, "the split-K hand-over must fit in the staging ring");
    constexpr int SMEM = ((PH >= 1 && PH <= 3) ? 4 * (BM + BN) * 64 : NS * ((BM + BN) * 128 + (PH == 5 ? 1024 : 0))) + (BM + BN) * 16 + BM * 4 + BN * 4 + (CONV ? BN * 4 : 0)      // staging ring + fused-LayerNorm block + the tile's bias (+ time-embedding row)
                       + (PH == 3 ? BM * f8_block_cap(BN) : 0);                                          // + the tile's MX block scales of A
    static_assert(SMEM <= 160 * 1024, "LDS");
    static bool attr_set = false;   // idempotent; racing threads set the same value
    auto kern = gemm_conv_kernel<BM, BN, WM, WN, NS, CONV, LW, PH, KS, CS, EK, SC>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
    p.group_m = BM >= 256 ? 4 : 8;
    if (BM >= 256 && BN >= 256) p.group_m = 8;
    // the kernels remap the LINEAR workgroup id over the whole (tiles, slices) grid in 32-bit arithmetic (common.h xcd_remap_grid)
    if ((int64_t)p.tiles_m * p.tiles_n * batch > 0x7fffffffLL) TMIX_FAIL(TMIX_ESHAPE, "gemm: %lld x %d workgroups exceed the 32-bit linear grid id", (long long)p.tiles_m * p.tiles_n, batch);
    dim3 grid(p.tiles_m * p.tiles_n, batch, 1);
    p.prof = tmix_prof_take(&p.prof_detail);
    tmix_prefetch_take(&p.pf, &p.pf_bytes);
    { const long long nthr = (long long)grid.x * grid.y * (LW ? LW : WM * WN * KS) * 64, lines = (p.pf_bytes + 127) >> 7;
      p.pf_per = p.pf ? (int)((lines + nthr - 1) / nthr) : 0; }
    kern<<<grid, (WM * WN * KS + LW) * 64, SMEM, st>>>(p);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}


// the instantiation of a tiling that holds what this launch needs and nothing else: the epilogue family EK (gemm_conv_kernel), with or without
// the column statistics of Params::cs_out
template <int BM, int BN, int WM, int WN, int NS, int CONV, int LW = 0, int PH = 0, int KS = 1, int CSOK = 1>
int launch_cs(Params& p, int batch, hipStream_t st) {
    const bool no_trans = p.n_trans_begin < 0;
    if constexpr (CONV == 1 && (LW == 0 || LW == 2) && PH < 4) {       // shortcut taps (validated: stride-1 conv, staged plain epilogue; bf16 only)
        if (p.S1) return p.cs_out ? launch_cfg<BM, BN, WM, WN, NS, CONV, LW, PH, KS, 1, 2, 1>(p, batch, st) : launch_cfg<BM, BN, WM, WN, NS, CONV, LW, PH, KS, 0, 2, 1>(p, batch, st);
    }
    if constexpr (CSOK) { if (p.cs_out) return launch_cfg<BM, BN, WM, WN, NS, CONV, LW, PH, KS, 1, 2>(p, batch, st); }         // (validated: plain staged epilogue)
    if constexpr (LW && PH >= 4) { if (p.f8copy) return launch_cfg<BM, BN, WM, WN, NS, CONV, LW, PH, KS, 0, 4>(p, batch, st); }   // (gemm_conv.hip:launch sends only straight-line launches here)
    if constexpr (!CONV) { if (no_trans && p.epilogue == TMIX_EPI_GEGLU && (p.wide & 2)) return launch_cfg<BM, BN, WM, WN, NS, CONV, LW, PH, KS, 0, 1>(p, batch, st); }
    if (no_trans && p.epilogue != TMIX_EPI_GEGLU && (p.wide & 1)) return launch_cfg<BM, BN, WM, WN, NS, CONV, LW, PH, KS, 0, 2>(p, batch, st);
    if constexpr (!CONV) { if (!no_trans && p.epilogue != TMIX_EPI_GEGLU && (p.wide & 1) && (p.wide & 4)) return launch_cfg<BM, BN, WM, WN, NS, CONV, LW, PH, KS, 0, 3>(p, batch, st); }
    return launch_cfg<BM, BN, WM, WN, NS, CONV, LW, PH, KS, 0, 0>(p, batch, st);
}

// one launcher per group of tilings (defined in gemm_inst_<g>.hip); returns -999 when `cfg` is not in the group
int launch_group0(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st);
int launch_group1(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st);
int launch_group2(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st);
int launch_group3(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st);
int launch_group4(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st);
int launch_group5(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st);      // fp8 in the lock-step loops (f8 = 3: per-row A scales, 4: MX blocks)
bool w22_eligible(const Params& p, int conv, int f8);                                   // gemm_w22.hip (tiling 23)
int launch_w22(Params& p, int batch, hipStream_t st, int l2_prefetcher = 0);            // (1: tiling 25, three DMA loaders + an L2 prefetcher wave)
bool ff1p_eligible(const Params& p, int conv, int f8, int batch);                        // gemm_ff1p.hip (tiling 24)
int launch_ff1p(Params& p, hipStream_t st);
bool convh_eligible(const Params& p, int conv, int f8);                                 // gemm_convh.hip (tiling 26)
int launch_convh(Params& p, hipStream_t st);
// gemm_qattn.hip: attn2.to_q + the cross-attention behind it in one launch (tmix_gemm_q_cross_attn)
struct QAExtra { const void* K; int64_t ldk, strideK; const void* Vt; int64_t ldvt, strideVt; void* O; int64_t ldo; int rows_per_image, Skv; float scale; };
int launch_qattn(Params& p, const QAExtra& x, int batch, hipStream_t st);

}  // namespace tmix_gemm
