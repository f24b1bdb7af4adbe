// gemm_convh.hip -- 3x3 NHWC convolution with the input HALO PATCH resident in LDS (TMIX_TILE_CONV_HALO = 26).
//
// Replaces the cuDNN convolutions of the SDXL ResnetBlock2D behind fusion_generation/fusion_sampling.py:340 (the UNet call), stride-1 form.
//
// The implicit-GEMM loop of gemm_kernel.h stages, for every one of the nine taps, the 128 x 64-channel slice of the SHIFTED input pixels of its M-tile: the
// same pixel travels L2 -> LDS nine times (and, tap-major, the re-reads are Cin / 64 K-tiles apart: through the fabric once the XCD's 4 MB L2 has turned over --
// rocprof FETCH_SIZE 396 MB per launch against 88 MB algorithmic).  Here an M-tile is a 4 x 32 block of output pixels, the K loop is
// CHANNEL-CHUNK major, and for every 64-channel chunk the (4 + 2) x (32 + 2) input patch is DMA'd ONCE into LDS (204 rows of 128 bytes, 1.6 x the tile's own
// pixels instead of 9 x); the nine taps are nine K-tiles whose A fragments are SHIFTED reads of that patch (lane's patch row + ky * 34 + kx), only the weights
// stream per tap.  LDS-DMA instructions per K-tile: 20 (W) + 28 / 9 (patch) = 23 instead of 36.
//   * tile 4 x 32, wave w owns output row w of the tile: a 32-row MFMA fragment is 32 CONSECUTIVE patch rows, so the row swizzle of gemm_kernel.h (16-byte chunk c
//     of LDS row r at position c ^ ((r >> 1) & 7)) stays conflict-free under every tap shift (the 16 rows a ds_read_b128 service group touches are distinct mod 16);
//   * four math waves (32 x 160 each, v_mfma_f32_32x32x16_bf16, the accumulator layout and epilogue arithmetic of tilings 12 / 20) + four loader waves, one per
//     SIMD, that issue every LDS-DMA with counted vmcnt waits; weight ring of four 20 KB stages, two patch buffers of 28 KB (the next chunk's patch lands while
//     this chunk's taps multiply);
//   * epilogue: the staged plain form -- bias, the image's time-embedding row, residual, bf16 store, GroupNorm column statistics (cs_out) -- rows mapped back
//     from the tile's 4 x 32 block to NHWC pixel order.
//   * shortcut taps (conv2 + conv_shortcut in one launch): behind the chunks the K loop walks 64-channel chunks of the block's input tensor(s) at the output
//     pixel as dense A tiles through a three-slot ring laid over the patch buffers.
// Requirements (gemm_conv.hip routes everything else to the other tilings): mode TMIX_CONV_S1, bf16, Wo % 32 == 0, Ho % 4 == 0, Cout % 160 == 0, Cin % 64 == 0,
// staged epilogue.  The accumulation order is chunk-major (the other tilings: tap-major): same products, another fp32 summation order.
//
// MFMA roofline: 2 * B * H * W * Cout * 9 * Cin flops per launch against the 2.5 PFLOP/s dense bf16 peak.
#include "gemm_kernel.h"

namespace tmix_gemm {

namespace {

constexpr int H_BN = 160, H_NS = 4, H_LW = 4, H_NW = 4;
constexpr int H_TR = 4, H_TC = 32;                       // output pixels of a tile: 4 rows x 32 columns = 128
constexpr int H_PC = H_TC + 2, H_PRV = (H_TR + 2) * H_PC; // patch pitch (34 pixels) and valid patch rows (204)
constexpr int H_PI = 28;                                  // LDS-DMA instructions per patch (8 rows of 128 bytes each): 224 rows, 7 per loader
constexpr int H_PATCH = 32 * 1024;                        // distance of the two patch buffers (28 KB used each: [0, 28K) and [32K, 60K))
constexpr int H_WT = H_BN * 128;                          // bytes of a weight stage (160 rows of 64 channels)
constexpr int H_WI = H_BN / 8;                            // LDS-DMA instructions per weight stage: 20, 5 per loader
constexpr int H_OFF_W = 2 * H_PATCH;                      // the weight ring sits behind the two patch buffers
// shortcut taps (conv2 + conv_shortcut of a ResnetBlock2D in one launch): behind the nine-tap chunks the K loop walks 64-channel chunks of the block's INPUT at
// the output pixel -- dense 128-row A tiles (16 KB, 4 LDS-DMA instructions per loader) through a ring of THREE slots laid over the patch region: with the last
// patch in buffer b the slots are (b ? 0 : 32K) + {0, 16K, 32K} mod 64K, i.e. the first two lie in the free buffer and land while the last chunk's taps multiply
constexpr int H_SA = 16 * 1024, H_LSA = 16 / H_LW;
constexpr int H_MAIN = H_OFF_W + H_NS * H_WT;
constexpr int H_LP = H_PI / H_LW, H_LWI = H_WI / H_LW;    // per loader: 7 patch, 5 weight instructions
constexpr int H_STG = 32 * (64 * 4 + 16);                 // epilogue staging patch per wave (32 rows x 64 fp32 columns, rows padded by 16 bytes)
static_assert(H_PI % H_LW == 0 && H_WI % H_LW == 0 && H_PI * 8 >= H_PRV && H_PI * 1024 <= H_PATCH - 4096, "loader geometry");
static_assert(2 * H_LWI + 2 * H_LSA <= 63 && 2 * H_LWI + H_LP <= 63, "vmcnt immediate");
static_assert(H_NW * H_STG <= H_NS * H_WT, "epilogue patches fit in the weight ring");

__global__ void __launch_bounds__((H_NW + H_LW) * 64, 2) conv_halo_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef TMIX_NO_KERNARG_TOUCH
    kernarg_touch<(int)sizeof(Params)>();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = w >= H_NW;
    const bool prof_on = p.prof != nullptr && tid == 0;
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0;
    if (prof_on) pt0 = prof_enter(p.prof, (blockIdx.x | blockIdx.y) == 0, p.prof_detail);
    // the NEXT launch's weights (tmix_gemm_prefetch_next): touched by the loader waves in front of their first DMA, whose counted waits cover the loads
    constexpr int PFU = 8;
    unsigned pf_keep[PFU];
#pragma unroll
    for (int u = 0; u < PFU; ++u) pf_keep[u] = 0;
    if (p.pf && loader) {
        const long long nwg = (long long)gridDim.x * gridDim.y, nth = H_LW * 64;
        const long long lines = (p.pf_bytes + 127) >> 7; const int per = p.pf_per;
        const long long first = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * nth + (tid - H_NW * 64);
#pragma unroll
        for (int u = 0; u < PFU; ++u) {
            const long long ln = first + (long long)u * nwg * nth;
            if (u < per && ln < lines) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_keep[u]) : "v"(p.pf + (ln << 7)) : "memory");
        }
    }
    // tile order: gemm_kernel.h's (an XCD owns a compact patch of group_m x (64 / group_m) tiles); m-tiles enumerate (image, tile row, tile column), column fastest,
    // so the group_m m-tiles of a group are neighbours along an image row and share their halo columns in the XCD's L2
    int bid, by;
    xcd_remap_grid(bid, by);
    const int per_group = p.group_m * p.tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * p.group_m;
    const int gsize = min(p.tiles_m - first_m, p.group_m);
    const int rem = bid - grp * per_group;
    const int tile_n = rem / gsize, tile_m = first_m + (rem - tile_n * gsize);
    const int n0 = tile_n * H_BN;
    const int txn = p.Wo / H_TC, tpi = (p.Ho / H_TR) * txn;          // tiles per image row, per image
    const int img = tile_m / tpi, trem = tile_m - img * tpi;
    const int ty = trem / txn, tx = trem - ty * txn;
    const int y0 = ty * H_TR, x0 = tx * H_TC;
    const int nchunks = p.Cin / BK;
    const int n1 = p.S1 ? p.c1s / BK : 0, nsc = n1 + (p.S2 ? p.c2s / BK : 0);          // shortcut K-tiles (first tensor, both)
    const int sa_base = ((nchunks - 1) & 1) ? 0 : H_PATCH;                             // slot j of the shortcut A ring: (sa_base + (j % 3) * 16K) mod 64K

    if (loader) {
        // ---- loader waves: LDS-DMA issue + counted waits only.  Loader s issues instructions g = 4 r + s of a patch (r < 7) and of a weight stage (r < 5); an
        // instruction covers 8 LDS rows of 128 bytes (lane -> row lane >> 3, 16-byte position lane & 7), and position q of row r holds source chunk q ^ ((r >> 1) & 7)
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.bytesA, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.bytesW, 0x00020000);
        const int s = w - H_NW, lrow = lane >> 3;
        unsigned poff[H_LP], woff[H_LWI];
#pragma unroll
        for (int r = 0; r < H_LP; ++r) {
            const int row = (r * H_LW + s) * 8 + lrow;                 // patch row = (py, px) of the (4 + 2) x (32 + 2) input window; rows >= 204 are padding
            const int py = row / H_PC, px = row - py * H_PC;
            const int iy = y0 + py - 1, ix = x0 + px - 1;
            const bool ok = (row < H_PRV) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd);
            const unsigned sw = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
            poff[r] = ok ? (unsigned)((img * p.H + iy) * p.Wd + ix) * (unsigned)p.Cin * 2u + sw : 0x80000000u;      // beyond num_records: zeros (the padding)
        }
#pragma unroll
        for (int r = 0; r < H_LWI; ++r) {
            const int row = (r * H_LW + s) * 8 + lrow;
            const unsigned sw = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
            woff[r] = (unsigned)(n0 + row) * (unsigned)p.ldw * 2u + sw;
        }
        // shortcut A tile: instruction g = 4 r + s covers dense rows 8 g .. 8 g + 7 = pixels (y0 + row / 32, x0 + row % 32) of the tile
        int spix[H_LSA]; unsigned ssw[H_LSA];
#pragma unroll
        for (int r = 0; r < H_LSA; ++r) {
            const int row = (r * H_LW + s) * 8 + lrow;
            spix[r] = (img * p.H + y0 + (row >> 5)) * p.Wd + x0 + (row & 31);
            ssw[r] = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
        }
        auto stage_sa = [&](int j) __attribute__((always_inline)) {       // shortcut K-tile j (wave-uniform): chunk j of S1, or chunk j - n1 of S2
            char* dst = smem + ((sa_base + (j % 3) * H_SA) & (2 * H_PATCH - 1));
            const bool first = j < n1;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? p.S1 : p.S2), 0, first ? p.bytesS1 : p.bytesS2, 0x00020000);
            const unsigned cs2 = (unsigned)(first ? p.c1s : p.c2s) * 2u, so = (unsigned)(first ? j : j - n1) * (BK * 2u);
#pragma unroll
            for (int r = 0; r < H_LSA; ++r) blds16(rs, j < nsc ? (unsigned)spix[r] * cs2 + ssw[r] : 0x80000000u, so, dst + (r * H_LW + s) * 1024);
        };
        auto stage_p = [&](int buf, int c, bool real) __attribute__((always_inline)) {
            char* dst = smem + buf * H_PATCH;
#pragma unroll
            for (int r = 0; r < H_LP; ++r) blds16(rsA, real ? poff[r] : 0x80000000u, (unsigned)c * (BK * 2), dst + (r * H_LW + s) * 1024);
        };
        // weight K-tile (chunk c, tap t): channels [64 c, 64 c + 64) of tap t of every output row -- OHWI rows of 9 * Cin elements
        // (c == nchunks: shortcut K-tile t -- the shortcut tensors' channels sit behind the nine taps in every weight row)
        auto stage_w = [&](int slot, int c, int t) __attribute__((always_inline)) {
            char* dst = smem + H_OFF_W + slot * H_WT;
#ifdef TMIX_ABL_WSEQ      // dev A/B builds only (wrong results, timing valid): the weight K-tiles read as if the rows were stored chunk-major, i.e. consecutive 128-byte pieces
            const unsigned so = (unsigned)(c * 9 + t) * (BK * 2u);
#else
            const unsigned so = c < nchunks ? (unsigned)(t * p.Cin + c * BK) * 2u : (unsigned)(9 * p.Cin + t * BK) * 2u;
#endif
#pragma unroll
            for (int r = 0; r < H_LWI; ++r) blds16(rsW, woff[r], so, dst + (r * H_LW + s) * 1024);
        };
        const int nk = nchunks * 9 + nsc;
        // prologue: patch 0 and weight tiles 0, 1 in front of the first barrier, tile 2 behind it (as tiling 21: the math waves start as soon as tile 0 is there)
        stage_p(0, 0, true);
        stage_w(0, 0, 0);
        stage_w(1, 0, 1);
        wait_vmcnt<H_LWI>();
#pragma unroll
        for (int u = 0; u < PFU; ++u) asm volatile("" :: "v"(pf_keep[u]));
        __builtin_amdgcn_s_barrier();
        stage_w(2, 0, 2);
        // K-tile kt = 9 c + t: weight tile kt + 3 goes to the ring slot the barrier of iteration kt - 1 released; at t == 0 the NEXT chunk's patch goes to the
        // other patch buffer (its last reader was chunk c - 1, whose last barrier is behind us; a dummy with out-of-range offsets behind the last chunk keeps the
        // instruction counts of the waits uniform).  The wait in front of the barrier that ends K-tile kt makes weight tile kt + 1 visible -- everything issued
        // before it included, i.e. the patch issued at t == 0 has landed by t == 3: long before chunk c + 1 reads it.
        int c3 = 0, t3 = 3, slot3 = 3;                       // (chunk, tap, ring slot) of weight tile kt + 3
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kt = c * 9 + t;
                const bool more = kt + 3 < nk;
                const bool sc_next = nsc > 0 && c == nchunks - 1;     // behind the last chunk come shortcut tiles: their first two A tiles instead of a patch
                if (more) stage_w(slot3, c3, t3);
                if (t == 0) { if (sc_next) { stage_sa(0); stage_sa(1); } else stage_p((c + 1) & 1, c + 1, c + 1 < nchunks); }
                if (more) { if (t <= 2) { if (sc_next) wait_vmcnt<2 * H_LWI + 2 * H_LSA>(); else wait_vmcnt<2 * H_LWI + H_LP>(); } else wait_vmcnt<2 * H_LWI>(); }
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                slot3 = (slot3 + 1) & 3;
                if (c3 < nchunks) { if (++t3 == 9) { t3 = 0; ++c3; } } else ++t3;
            }
        }
        // shortcut K-tiles: tile j reads weight tile nk_main + j and A slot j % 3; A tile j + 2 goes to the slot tile j - 1 read (its barrier is behind us).  The wait
        // makes weight tile kt + 1 AND A tile j + 1 (issued one iteration ago, behind weight tile kt + 2) visible: only what this iteration issued may stay in flight
        for (int j = 0; j < nsc; ++j) {
            const bool more_w = j + 3 < nsc, more_a = j + 2 < nsc;
            if (more_w) stage_w(slot3, c3, t3);
            if (more_a) stage_sa(j + 2);
            if (more_w) wait_vmcnt<H_LWI + H_LSA>(); else if (more_a) wait_vmcnt<H_LSA>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            slot3 = (slot3 + 1) & 3;
            ++t3;
        }
        return;
    }

    // ---------------------------------------------------------------- math waves: wave w = output row y0 + w of the tile, 32 pixels x 160 output channels
    const int l31 = lane & 31, lhi = lane >> 5;
    float* bias_lds = (float*)(smem + H_MAIN);
    float* rgb_lds = bias_lds + H_BN;
    if (tid < H_BN) {
        bias_lds[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
        // the image's time-embedding row (conv1 of a ResnetBlock2D); a tile lies inside one image
        const int64_t m_first = (int64_t)img * p.Ho * p.Wo;
        rgb_lds[tid] = p.rgb ? p.rgb[(m_first / p.rows_per_group) * p.N + n0 + tid] : 0.f;
    }
    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // A fragment of (patch buffer, tap, k-step kk): patch row pp = (w + ky) * 34 + l31 + kx, source chunk 2 kk + lhi at position chunk ^ ((pp >> 1) & 7)
    // (ppv is pinned opaque at every tap -- asm below --, or the compiler hoists the 36 lane offsets of the (tap, k-step) pairs out of the chunk loop and spills)
    int ppv = w * H_PC + l31;
    auto rd_a = [&](int buf, int t, int kk) __attribute__((always_inline)) -> frag_ab {
        const int pp = ppv + (t / 3) * H_PC + (t % 3);
        return *(const frag_ab*)(smem + buf * H_PATCH + pp * 128 + ((((kk * 2) + lhi) ^ ((pp >> 1) & 7)) << 4));
    };
    // W fragment j of (ring slot, k-step kk): row 32 j + l31
    const int fsw = (lane >> 1) & 7;
    const int offW = H_OFF_W + l31 * 128;
    auto rd_b = [&](int slot, int j, int kk) __attribute__((always_inline)) -> frag_ab {
        return *(const frag_ab*)(smem + offW + slot * H_WT + j * 32 * 128 + ((((kk * 2) + lhi) ^ fsw) << 4));
    };
    // residual rows in the read-back layout of the epilogue, requested in front of the last K-tile: two 64-column chunks (8 lanes per row, 4 passes of 8 rows)
    // and the trailing 32-column chunk (4 lanes per row, 2 passes of 16 rows); pixel row r of the wave's block is NHWC row mrow0 + r
    const int64_t mrow0 = ((int64_t)img * p.Ho + y0 + w) * p.Wo + x0;
    uint4 rw[10];
    auto epi_prefetch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 10; ++q) rw[q] = make_uint4(0u, 0u, 0u, 0u);
        if (p.R) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ps = 0; ps < 4; ++ps)
                    rw[c * 4 + ps] = *(const uint4*)(p.R + (mrow0 + ps * 8 + (lane >> 3)) * p.ldr + n0 + c * 64 + (lane & 7) * 8);
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
                rw[8 + ps] = *(const uint4*)(p.R + (mrow0 + ps * 16 + (lane >> 2)) * p.ldr + n0 + 128 + (lane & 3) * 8);
        }
    };

    __builtin_amdgcn_s_barrier();                      // patch 0 and weight tile 0 have landed
    asm volatile("" ::: "memory");
    if (prof_on) pt1 = prof_now();
    frag_ab fa[2], fb[2][5];
    fa[0] = rd_a(0, 0, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) fb[0][j] = rd_b(0, j, 0);
    // one k-step (16 of the K-tile's 64 channels): five MFMAs on register set S with the six fragment reads of the NEXT k-step (patch buffer nb, tap nt, ring
    // slot ns, k-step nkk; into set 1 - S) behind the first of them -- issue order pinned
    auto kstep = [&](const int S, const int nb, const int nt, const int ns, const int nkk) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[S][j], fa[S], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifdef TMIX_HALO_ABL_LDS      // dev A/B builds only (wrong results): 16 instead of 24 fragment reads per K-tile -- is the loop bound by LDS read bytes?
            if (j == 0) fa[1 - S] = rd_a(nb, nt, nkk);
            if (j < 3) fb[1 - S][j] = rd_b(ns, j, nkk); else fb[1 - S][j] = fb[1 - S][j - 3];
#elif defined(TMIX_HALO_READS_EVEN)     // dev A/B builds: one W fragment behind every MFMA (the first form)
            if (j == 0) fa[1 - S] = rd_a(nb, nt, nkk);
            fb[1 - S][j] = rd_b(ns, j, nkk);
#else
            // the six reads of the next k-step behind the FIRST three MFMAs (two each, as the lock-step loops of gemm_kernel.h deal them): the last one is then two
            // MFMAs old when the K-tile's hand-over waits for lgkmcnt(0), instead of zero
            if (j == 0) { fa[1 - S] = rd_a(nb, nt, nkk); fb[1 - S][0] = rd_b(ns, 0, nkk); }
            else if (j == 1) { fb[1 - S][1] = rd_b(ns, 1, nkk); fb[1 - S][2] = rd_b(ns, 2, nkk); }
            else if (j == 2) { fb[1 - S][3] = rd_b(ns, 3, nkk); fb[1 - S][4] = rd_b(ns, 4, nkk); }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int cur = 0;
#ifdef TMIX_HALO_PRIO         // dev A/B builds: the math waves above the loader wave of their SIMD in the issue arbiter
    __builtin_amdgcn_s_setprio(2);
#endif
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t == 8 && c == nchunks - 1) { epi_prefetch(); __builtin_amdgcn_sched_barrier(0); }
            asm volatile("" : "+v"(ppv));
            kstep(0, buf, t, cur, 1);
            kstep(1, buf, t, cur, 2);
            kstep(0, buf, t, cur, 3);
            // every fragment of this K-tile is in registers (its ring slot may be restaged behind the barrier); weight tile kt + 1 has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            cur = (cur + 1) & 3;
            // (behind the last K-tile these reads fetch stale LDS into registers nobody uses)
            kstep(1, t == 8 ? (buf ^ 1) : buf, t == 8 ? 0 : t + 1, cur, 0);
        }
    }
    if (nsc > 0) {
        // ---- shortcut K-tiles: dense A tiles (row w * 32 + l31 of the slot, the swizzle of the W rows), same weight ring
        const int offS = (w * 32 + l31) * 128;
        auto rd_as = [&](int j, int kk) __attribute__((always_inline)) -> frag_ab {
            return *(const frag_ab*)(smem + ((sa_base + (j % 3) * H_SA) & (2 * H_PATCH - 1)) + offS + ((((kk * 2) + lhi) ^ fsw) << 4));
        };
        auto kstep_s = [&](const int S, const int nj, const int ns, const int nkk) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[S][j], fa[S], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) { fa[1 - S] = rd_as(nj, nkk); fb[1 - S][0] = rd_b(ns, 0, nkk); }
                else if (j == 1) { fb[1 - S][1] = rd_b(ns, 1, nkk); fb[1 - S][2] = rd_b(ns, 2, nkk); }
                else if (j == 2) { fb[1 - S][3] = rd_b(ns, 3, nkk); fb[1 - S][4] = rd_b(ns, 4, nkk); }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        fa[0] = rd_as(0, 0);                               // (the main loop's last k-step fetched a stale patch row here; the W fragments of tile 0 are already right)
        for (int j = 0; j < nsc; ++j) {
            kstep_s(0, j, cur, 1);
            kstep_s(1, j, cur, 2);
            kstep_s(0, j, cur, 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            cur = (cur + 1) & 3;
            kstep_s(1, j + 1, cur, 0);
        }
    }
    if (prof_on) pt2 = prof_now();

    // ---- staged epilogue (the straight-line form of gemm_kernel.h, FM = 1, FN = 5).  32 x 32 accumulator: lane holds pixel l31, output channels
    // 8 g + 4 lhi + r of fragment j (register 4 g + r).  Per chunk the wave parks 32 pixels x 64 (32) fp32 columns in its LDS patch and reads them back
    // row-major: 8 (4) lanes x 8 columns per pixel, 16-byte residual loads (prefetched) and C stores.
    char* stg = smem + H_OFF_W + w * H_STG;
    const int64_t blk = mrow0 >> 5;                    // the 32 pixels of this wave are one 32-row block of the NHWC matrix (x0 and Wo are multiples of 32): cs_out[M / 32][2][N] as every tiling writes it
    auto chunk = [&](int j0, auto cf_tag) __attribute__((always_inline)) {
        constexpr int CF = decltype(cf_tag)::value, CW = CF * 32, SR = CW * 4 + 16, LPR = CW / 8, RPI = 64 / LPR, NP = 32 / RPI;
        const int rr = lane / LPR, cc = (lane % LPR) * 8;
        const int nl = j0 * 32 + cc, nc = n0 + nl;
        const float4 bq0 = *(const float4*)(bias_lds + nl), bq1 = *(const float4*)(bias_lds + nl + 4);
        const float4 t0 = *(const float4*)(rgb_lds + nl), t1 = *(const float4*)(rgb_lds + nl + 4);
        const float bq[8] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w};
        const float tq[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        float cv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) cv[k] = 0.f;
#pragma unroll
        for (int jj = 0; jj < CF; ++jj)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(stg + l31 * SR + (jj * 32 + g * 8 + lhi * 4) * 4) =
                    make_float4(acc[j0 + jj][g * 4 + 0], acc[j0 + jj][g * 4 + 1], acc[j0 + jj][g * 4 + 2], acc[j0 + jj][g * 4 + 3]);
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int r = ps * RPI + rr;
            const float4 v0 = *(const float4*)(stg + r * SR + cc * 4), v1 = *(const float4*)(stg + r * SR + cc * 4 + 16);
            float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fmaf(o[k], 1.f, bq[k]) + tq[k];
            const uint4 rq = rw[(CF == 2 ? (j0 / 2) * 4 : 8) + ps];
            const unsigned ru[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[2 * k] += __uint_as_float(ru[k] << 16); o[2 * k + 1] += __uint_as_float(ru[k] & 0xffff0000u); }
            uint4 v;
            v.x = pack_bf2(o[0], o[1]); v.y = pack_bf2(o[2], o[3]); v.z = pack_bf2(o[4], o[5]); v.w = pack_bf2(o[6], o[7]);
            *(uint4*)(p.C + (mrow0 + r) * p.ldc + nc) = v;
            if (p.cs_out) {                               // column statistics of the values AS STORED
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(u[k] << 16), hi = __uint_as_float(u[k] & 0xffff0000u);
                    cv[2 * k] += lo; cv[2 * k + 1] += hi;
                    cv[8 + 2 * k] = fmaf(lo, lo, cv[8 + 2 * k]); cv[9 + 2 * k] = fmaf(hi, hi, cv[9 + 2 * k]);
                }
            }
        }
        if (p.cs_out) {
            // the 8 (16) lanes that share this lane's columns are reduced as a reduce-scatter (gemm_kernel.h cs_flush): lane (bit 5 = statistic, bit 4 = column half)
            // ends with four adjacent columns of one plane
            float tt[8], u[4];
#pragma unroll
            for (int k = 0; k < 8; ++k) tt[k] = swap32_sum(cv[k], cv[8 + k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = swap16_sum(tt[k], tt[k + 4]);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] += dpp_row<0x128>(u[k]);
            if constexpr (CF == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) u[k] += dpp_row<0x124>(u[k]);
            }
            if (!(lane & (CF == 1 ? 12 : 8)))
                *(float4*)(p.cs_out + (blk * 2 + (lane >> 5)) * p.N + nc + ((lane >> 4) & 1) * 4) = make_float4(u[0], u[1], u[2], u[3]);
        }
    };
    chunk(0, std::integral_constant<int, 2>{});
    chunk(2, std::integral_constant<int, 2>{});
    chunk(4, std::integral_constant<int, 1>{});
    if (prof_on) prof_leave(p.prof, p.prof_detail, pt0, pt1, pt2);
}

}  // namespace

// can the halo-patch kernel run this convolution?
bool convh_eligible(const Params& p, int conv, int f8) {
    return conv && !f8 && p.mode == TMIX_CONV_S1 && p.ntaps == 9 && !p.scaleA && (p.wide & 1) && (p.Wo % H_TC) == 0 && (p.Ho % H_TR) == 0 && (p.N % H_BN) == 0
           && (p.Cin % BK) == 0 && (!p.rgb || p.rows_per_group % (p.Ho * p.Wo) == 0) && (!p.R || ((p.ldr % 8) == 0 && aligned16(p.R)));
}

int launch_convh(Params& p, hipStream_t st) {
    constexpr int SMEM = H_MAIN + 2 * H_BN * 4;
    static_assert(SMEM <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) TMIX_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    p.tiles_m = p.M / 128; p.tiles_n = p.N / H_BN;
    p.group_m = 8;
    if ((int64_t)p.tiles_m * p.tiles_n > 0x7fffffffLL) TMIX_FAIL(TMIX_ESHAPE, "conv3x3: %lld workgroups exceed the 32-bit linear grid id", (long long)p.tiles_m * p.tiles_n);
    dim3 grid(p.tiles_m * p.tiles_n, 1, 1);
    p.prof = tmix_prof_take(&p.prof_detail);
    tmix_prefetch_take(&p.pf, &p.pf_bytes);
    { const long long nthr = (long long)grid.x * H_LW * 64, lines = (p.pf_bytes + 127) >> 7;
      p.pf_per = p.pf ? (int)((lines + nthr - 1) / nthr) : 0; }
    conv_halo_kernel<<<grid, (H_NW + H_LW) * 64, SMEM, st>>>(p);
    TMIX_LAUNCH_CHECK();
    return TMIX_OK;
}

}  // namespace tmix_gemm
