/* tmix.h -- C ABI of libtmix_hip.so, the MI355X (gfx950) kernels behind the Tweedie-mix
 * denoising loop.
 *
 * The reference (KwonGihyun/TweedieMix) is pure Python and has no FFI of its own: its seams are
 * the Python call sites listed below.  Each entry point here replaces the arithmetic behind one of
 * those call sites; tweediemix_amd/ binds them with ctypes (see INTEGRATION.md for the stub a
 * maintainer of the reference would add).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  Every pointer is a BORROWED device pointer (the caller
 *    -- PyTorch's allocator in our host code -- owns the memory).  Nothing here allocates or frees.
 *  - every launch function takes the hipStream_t to enqueue on (as void*), is asynchronous, never
 *    synchronises, and is capturable into a hipGraph.
 *  - return value: 0 = ok, <0 = TMIX_E* argument error (nothing was launched), >0 = hipError_t.
 *    tmix_last_error_string() gives a thread-local description of the last non-zero return.
 *  - bf16 = raw uint16 storage, row-major.  Activations are NHWC ([B, H*W, C]); Linear weights are
 *    torch layout [N_out, K_in]; conv weights are OHWI [C_out, 3, 3, C_in].
 */
#ifndef TMIX_H
#define TMIX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMIX_VERSION 100

enum { TMIX_OK = 0, TMIX_EINVAL = -1, TMIX_ESHAPE = -2, TMIX_EARCH = -3, TMIX_EALIGN = -4 };
enum { TMIX_F32 = 0, TMIX_F16 = 1, TMIX_BF16 = 2 };

int tmix_version(void);
const char* tmix_last_error_string(void);
/* 0 if the current device is gfx950, TMIX_EARCH otherwise, >0 on HIP errors. */
int tmix_check_device(void);
/* The library's debug / A-B environment switches (TMIX_GN_NO_SMALL, TMIX_ATTN_NO_SPLIT, TMIX_ATTN_GENERAL, TMIX_NARROW_EPILOGUE) are read ONCE per process, on first
 * use, so that a launch, the query that accounts for it (tmix_groupnorm_nhwc_launches, tmix_attn_split_ws_bytes) and a captured graph all see one value.  This re-reads
 * them (a test that flips a switch mid-process).  No reference counterpart. */
void tmix_env_refresh(void);

/* In-situ launch timing (measurement only; no reference counterpart).  Between tmix_prof_begin and tmix_prof_end every
 * tmix_gemm_bf16 / tmix_conv3x3_nhwc / tmix_attn_fwd / tmix_groupnorm_nhwc launch issued by THIS host thread -- also while
 * it is being captured into a hipGraph -- takes the next 8-word slot of `slots` (device memory, capacity slots) and records
 * into it, in ticks of the 100 MHz realtime clock: {start, end, sum(first operands landed - start), sum(main loop done -
 * start), sum(workgroup end - start), workgroups, -, -}.  detail = 0: start is stored by the first workgroup, end by every
 * workgroup (the last store wins), nothing else -- negligible cost, accurate to a store latency; detail = 1 (GEMM / conv /
 * attention): exact min / max and the per-workgroup phase sums through atomics (this perturbs launches with thousands of
 * workgroups).  The caller initialises every slot to {UINT64_MAX, 0, 0, 0, 0, 0, 0, 0} before each measured run / replay.
 * tmix_prof_end returns the number of slots handed out.  Launches outside such a bracket carry no instrumentation. */
int tmix_prof_begin(uint64_t* slots, int capacity, int detail);
int tmix_prof_end(void);

/* ---------------------------------------------------------------------------------------------
 * Fused CFG + Tweedie x0 + mask blend + DDIM update.
 * Replaces fusion_generation/fusion_sampling.py:376-385,430,471-472 (fusion branch),
 * :392-403 (resampling, first half), :406-412 / :424-430 / :436-447 (plain CFG / jumping).
 *   x        [n]            fp32 latent (one sample, n = C*h*w)
 *   eps      [rows, n]      UNet output, dtype eps_dtype (row 0 = unconditional)
 *   masks    [K, hw]        fp32 blend weights (FUSION only; broadcast over the C channels)
 *   out_x    [n]            fp32 next latent  (x0 itself when is_last)
 *   out_x0   [n] or NULL    fp32 Tweedie estimate
 * modes: FUSION   rows=K+1 : x0 = sum_c m_c * T(x, cfg(e0,e_{1+c}))
 *        PLAIN    rows=2   : x0 = T(x, cfg(e0,e1))
 *        RESAMPLE rows=K+1 : x0 = (K-1)*T(x,cfg(e0,e1)) - sum_{c<K-1} T(x,cfg(e0,e_{2+c}))
 *   with T(x,e) = (x - s1*e)/sa and out = sa_next*x0 + s1_next*e0.
 * sa=sqrt(at), s1=sqrt(1-at), *_next likewise for the target timestep (computed by the host in fp32).
 * eps_dtype F16 reproduces the reference's autocast rounding points (CFG combine and s1*e in
 * fp16, everything else fp32); BF16/F32 eps are combined in fp32.
 */
enum { TMIX_STEP_FUSION = 0, TMIX_STEP_PLAIN = 1, TMIX_STEP_RESAMPLE = 2 };
int tmix_fused_tweedie_step(const float* x, const void* eps, int eps_dtype, const float* masks,
                            float* out_x, float* out_x0, int K, int channels, int64_t hw, int mode,
                            float g, float sa, float s1, float sa_next, float s1_next, int is_last,
                            void* stream);
/* The same step with its coefficients in DEVICE memory, for hipGraph capture of a whole denoising step (the loop body of
 * fusion_sampling.py:490-494 becomes one graph replay per timestep): params = fp32 {t, sa, s1, sa_next, s1_next,
 * is_last (0/1), g, -}.  `seeds` co-batched trajectories run in one launch: x / out_x / out_x0 are [seeds][n], eps is
 * [seeds][rows][n] (rows = the UNet batch rows per seed), masks [seeds][K][hw] with mask_seed_stride = K*hw (0: one mask
 * set shared by all seeds).  x may alias out_x (in-place update of the latent state).  Bit-identical to the scalar form. */
int tmix_fused_tweedie_step_dev(const float* x, const void* eps, int eps_dtype, const float* masks,
                                int64_t mask_seed_stride, float* out_x, float* out_x0, int K, int channels,
                                int64_t hw, int mode, int rows, int seeds, const float* params, void* stream);
/* Head of the captured step (fusion_sampling.py:324-327, `latent_model_input = torch.cat([x] * rows)`, and the timestep
 * argument of the UNet call :340): latent[(s*rows + r)][n] = x[s][n] for every row r, t_dev[s*rows + r] = params[0]. */
int tmix_step_prologue(const float* x, float* latent, float* t_dev, const float* params, int seeds, int rows,
                       int64_t n, void* stream);

/* Video sampler (I2VGen-XL loop, BASELINE config #5; replaces video_gen/pipeline_i2vgen_xl.py:699-719):
 *   x [n], v [2n] (uncond rows first), out [n], all of `dtype` (TMIX_F32 / TMIX_F16 / TMIX_BF16); sa = sqrt(alpha(t)) etc. with
 *   the UN-shifted table of that pipeline (:480-482).  v' = v_u + g (v_t - v_u); eps = sa v' + s1 x; x0 = sa x - s1 v';
 *   out = sa_next x0 + s1_next eps.  TMIX_F16 rounds after every binary op like the reference's fp16 tensors. */
int tmix_vpred_step(const void* x, const void* v, void* out, int dtype, int64_t n, float g,
                    float sa, float s1, float sa_next, float s1_next, void* stream);
/* First-frame feature injection of the patched ResnetBlock2D.forward (video_gen/utils_attn.py:433-455), in place on
 * x [clips][frames][per_frame]: frames 1.. <- frame 0 (hard) or interp*frame0 + one_minus_interp*frame. */
int tmix_frame_inject(void* x, int dtype, int clips, int frames, int64_t per_frame, int hard, float interp,
                      float one_minus_interp, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogues:  C[b] = epi(A[b] (MxK) * W[b]^T (NxK)).
 * Replaces the Linear layers inside the UNet call at fusion_sampling.py:340/374/406/414/440
 * (diffusers Attention.to_q/to_k/to_v/to_out, FeedForward, proj_in/out, conv_shortcut) and the
 * per-row concept weights of utils_custom.py:64-83 / utils_lora.py:65-79,113-119 (batch with
 * strideW != 0 selects one weight set per batch row).
 * Requirements: K % 64 == 0, lda/ldw % 8 == 0, 16-byte aligned pointers.
 */
enum { TMIX_EPI_NONE = 0, TMIX_EPI_GEGLU = 1, TMIX_EPI_F32OUT = 2 /* C is fp32 [M][ldc] (attention scores of the VAE) */,
       TMIX_EPI_GELU = 3 /* gelu(acc + bias), erf form (OpenCLIP-bigG MLP) */,
       TMIX_EPI_QUICKGELU = 4 /* x * sigmoid(1.702 x) (CLIP-L MLP) */ };
/* workgroup tilings of the MFMA mainloop (BM x BN, waves, LDS ring depth); AUTO = built-in heuristic */
enum { TMIX_TILE_AUTO = 0, TMIX_TILE_128x128_S2 = 1, TMIX_TILE_256x128_S3 = 2, TMIX_TILE_128x128_S4 = 3,
       TMIX_TILE_256x256_S2 = 4, TMIX_TILE_256x128_W4 = 5, TMIX_TILE_256x256_W4 = 6, TMIX_TILE_128x160_S2 = 7,
       /* the same tilings with one extra LOADER wave that issues every LDS-DMA (the math waves only read LDS and issue MFMAs) */
       TMIX_TILE_128x160_S2_LW = 8, TMIX_TILE_256x128_S3_LW = 9, TMIX_TILE_128x128_S2_LW = 10, TMIX_TILE_256x256_S2_LW = 11,
       TMIX_TILE_128x160_S4 = 12 /* tiling 7 with a 4-deep LDS ring (one workgroup per CU) */,
       TMIX_TILE_64x160_W5 = 13 /* five waves of 64x32, 4-deep ring: 2048 x 1280 is exactly 256 tiles (one per CU) */,
       TMIX_TILE_256x320_S2 = 14 /* eight waves of 64x160: 2048 x 10240 (GEGLU up-projection) is exactly 256 tiles */,
       TMIX_TILE_32x160_W5 = 15 /* five waves of 32x32: 1024 x 1280 (one batch row per chain) is 256 tiles */,
       TMIX_TILE_256x256_PH = 16 /* eight waves, K slices of 32 through a four-slot ring, the second wave of each SIMD one barrier behind the first */,
       TMIX_TILE_256x128_PH = 17,
       TMIX_TILE_128x160_S4_K2 = 18 /* tiling 12 with in-workgroup split-K: two groups of four waves share the tile (eight waves stage) */,
       TMIX_TILE_128x160_S3_LW = 19 /* 128x160, 3-deep ring, plus ONE loader wave: the four math waves issue no VMEM instruction in the K loop */,
       TMIX_TILE_128x160_S4_LW2 = 20 /* tiling 12 plus TWO loader waves (each issues every other LDS-DMA instruction of a K-tile) */,
       TMIX_TILE_128x160_S4_LW4 = 21 /* tiling 12 plus FOUR loader waves: one per SIMD, nine LDS-DMA instructions of a K-tile each */,
       TMIX_TILE_256x320_PH = 22 /* 256x320 with the phase-offset mainloop (eight waves of 64x160): bf16 GEMM, no transposed region */,
       TMIX_TILE_128x160_W22 = 23 /* 128x160 over 2 x 2 math waves of 64x80 (16x16x32 MFMA) + four loader waves: 72 KB instead of 96 KB of LDS fragment reads per K-tile;
                                      plain staged bf16 epilogue (bias, folded LayerNorm, residual, row statistics) -- other launches run as tiling 21 */,
       /* ids 24 and 25 are RESERVED: two measured-and-rejected experiments (256x320 on persistent workgroups; tiling 23 with an L2 prefetcher wave) that only dev
          builds contain (make EXPERIMENTAL=1, tools/build_variant.sh); the shipped library runs them as tilings 14 and 23, whose bits they reproduce */
       TMIX_TILE_CONV_HALO = 26 /* tmix_conv3x3_nhwc only: stride-1 3x3 convolution on 4 x 32 pixel tiles with the (4 + 2) x (32 + 2) input patch of every 64-channel chunk resident
                                   in LDS -- the nine taps are shifted fragment reads, the input travels L2 -> LDS 1.6 x instead of 9 x.  W %% 32 == 0, H %% 4 == 0, Cout %% 160 == 0, no shortcut
                                   taps; other launches run as tiling 20 (a GEMM: 21).  Channel-chunk-major accumulation order (the other tilings: tap-major) */,
       TMIX_TILE_COUNT = 26 };
typedef struct {
    const void* A;  int64_t lda, strideA;        /* bf16 [batch][M][lda]                          */
    const void* W;  int64_t ldw, strideW;        /* bf16 [batch|1][N][ldw]                        */
    void*       C;  int64_t ldc, strideC;        /* bf16 [batch][M][ldc] (N/2 cols when GEGLU)    */
    const float* bias;  int64_t strideBias;      /* fp32 [N] or NULL                              */
    const void* residual; int64_t ldr, strideR;  /* bf16 [batch][M][ldr] or NULL (added in fp32)  */
    const float* rowgroup_bias;                  /* fp32 [M/rows_per_group][N] or NULL            */
    int32_t rows_per_group;
    void*   Ct; int64_t ldct, strideCt;          /* transposed output for columns >= n_trans_begin */
    int32_t n_trans_begin;                       /*   Ct[b][n - n_trans_begin][m]; <0 = none       */
    int32_t M, N, K, batch;
    int32_t epilogue;
    int32_t tile_cfg;                            /* TMIX_TILE_AUTO or a specific tiling (autotuned by the host) */
    /* --- LayerNorm fused across the GEMM pair that surrounds it (diffusers BasicTransformerBlock.norm1/2/3):
     * producer: row_stats_out[b][t][m] = {sum, sum of squares} over the columns of column-tile t of the STORED
     *   (bf16-rounded) output row m: plain stores in a fixed reduction order (no atomics, nothing to zero, replays are
     *   bit-reproducible).  t < tmix_gemm_stats_parts(N, tile_cfg); tile_cfg must be explicit (not TMIX_TILE_AUTO).
     * consumer: A holds the raw (un-normalised) rows and ln_stats the producer's ln_parts partials per row; with
     *   W' = W*gamma (column-scaled), ln_colsum[n] = sum_k W'[n][k] and bias[n] = sum_k W[n][k]*beta[k] (+ the layer's
     *   bias) the kernel forms  rstd[m] * (acc[m][n] - mean[m] * ln_colsum[n]) + bias[n]  ==  Linear(LayerNorm(x)).
     * Statistics live as fp32 {sum, sumsq} pairs [parts][ld rows]; batch b of a batched GEMM starts stride floats in
     * (so a [B*S]-row buffer serves both the flat M = B*S GEMMs and the batched M = S ones: ld = B*S, stride = 2*S). */
    float* row_stats_out; int64_t strideStatsOut, ldStatsOut;     /* NULL or fp32 [parts][ldStatsOut][2] */
    const float* ln_stats; int64_t strideLnStats, ldLnStats;      /* NULL or fp32 [ln_parts][ldLnStats][2] */
    const float* ln_colsum; int64_t strideLnColsum;   /* fp32 [batch?][N] (stride 0: shared) */
    float ln_inv_c, ln_eps;                           /* 1/C and eps of the LayerNorm */
    int32_t ln_parts, reserved0;                      /* partials per row in ln_stats, 1..16 */
    /* --- GroupNorm statistics from the launch that WRITES the normalised tensor (diffusers ResnetBlock2D.norm1/norm2,
     * Transformer2DModel.norm, conv_norm_out read what a conv / proj_out / conv_shortcut launch has just stored): NULL, or fp32
     * [M / TMIX_COLSTATS_ROWS][2][N]: plane 0 the sum, plane 1 the sum of squares of every column over rows [32 r, 32 r + 32) of
     * the STORED (bf16-rounded) output.  Plain stores in a fixed order (nothing to zero, replays are bit-reproducible), the same
     * layout under every tiling; tmix_groupnorm_nhwc_pre consumes it.  Plain bf16 epilogue only (no transposed region, no
     * activation, no row_stats_out / e4m3 copy), batch == 1, M %% 32 == 0, N %% 8 == 0, 16-byte aligned C / residual rows. */
    float* col_stats_out;
    /* --- periodic weight sets (several independent seeds share every UNet launch: batch rows [seed][concept], and row b of the batch wants the
     * weights of concept b %% w_period): 0 = W / bias / ln_colsum (and the fp8 W scales) are indexed by the batch slice as their strides say;
     * P > 0 (batch %% P == 0) = slice b reads set b %% P of P stored sets -- no gathered per-row copies of the merged LoRA weights
     * (utils_lora.py:65-79,113-119 applied per row), and the slices that share a set are issued back to back. */
    int32_t w_period, reserved1;
} tmix_gemm_desc;
int tmix_gemm_bf16(const tmix_gemm_desc* d, void* stream);
/* Hint for the NEXT tmix_gemm_bf16 / tmix_conv3x3_nhwc launch issued by this host thread (consumed and cleared by it): while its
 * workgroups wait for their own first operands they touch [next_weights, next_weights + bytes) -- one 4-byte load per 128-byte line, the
 * range split evenly over the grid -- so that the weights of the launch that FOLLOWS it are in the memory-side (Infinity) cache when that
 * launch starts (a dependent chain of launches otherwise meets every weight cold from HBM: fusion_sampling.py:340 calls the UNet's
 * ~490 Linear / Conv2d layers back to back).  Purely a performance hint: no effect on results; bytes <= 0 or NULL clears it.
 * The touches share the hinting launch's memory queues: from a launch of tens of microseconds name a few MB of a large tensor rather than all of it.
 * `stream` is ignored (present so that the call has the shape of every other launch entry). */
int tmix_gemm_prefetch_next(const void* next_weights, int64_t bytes, void* stream);
/* The same GEMM on OCP fp8 (e4m3) operands -- the reference's precision bar is fp16 autocast (fusion_sampling.py:492); this is
 * the optional lower-precision path for the FF / QKV projections (`--dtype fp8`), never the default.  A and W hold e4m3 BYTES
 * (lda / ldw / strides in elements = bytes, multiples of 16; K %% 64 == 0) and every A row and W row carries one power-of-two
 * scale, an E8M0 exponent byte: A[m][k] = a8[m][k] * 2^(scale_a[m] - 127).  scale_a is [batch][M], scale_w [N] (or [batch][N]
 * when strideW != 0).  v_mfma_scale_f32_32x32x64_f8f6f4 applies both scales in hardware (per-row instead of the MX format's
 * per-32 blocks, so a lane keeps its scales for the whole K loop); accumulation and every epilogue of tmix_gemm_bf16 (bias,
 * fused LayerNorm on A, GEGLU, residual, transposed V, row statistics) are unchanged.  tile_cfg: TMIX_TILE_AUTO,
 * TMIX_TILE_256x256_PH or TMIX_TILE_256x128_PH.
 * Two flags in tmix_gemm_desc.reserved0 chain fp8 GEMMs without a quantiser pass in between (FF up-projection -> down-projection):
 *   TMIX_F8_GEGLU_OUT       with TMIX_EPI_GEGLU: C receives e4m3 BYTES [M][N/2] (ldc in bytes) and Ct (ldct >= batch*M) the E8M0 scales of
 *                           every 32 output columns, k-block major: Ct[(col / 32) * ldct + b*M + m] -- the MX block form;
 *   TMIX_F8_A_BLOCK_SCALES  scale_a is such a block-scale array [K/32][batch*M] instead of one byte per row (the instruction applies
 *                           a lane's scale to exactly its 32 K values); scale_w stays per row.  The K/32 blocks of a tile stay in
 *                           LDS for the whole K loop: K <= 7168 (256x256 tiles up to K = 2816, 256x128 beyond), M %% 4 == 0. */
/*   TMIX_F8_COPY_OUT        (tmix_gemm_bf16 and tmix_gemm_fp8, plain epilogue, no transposed region): besides the bf16 rows in C the launch
 *                           leaves their e4m3 copy with MX block scales -- what a quantiser pass over C would produce -- for the next GEMM
 *                           that reads C as its A operand: Ct = bytes [batch*M][ldct] (ldct >= N), and the scale array [N/32][batch*M]
 *                           starts strideCt BYTES behind Ct.  M %% 32 == 0, N %% 32 == 0. */
enum { TMIX_F8_A_BLOCK_SCALES = 1, TMIX_F8_GEGLU_OUT = 2, TMIX_F8_COPY_OUT = 4 };
#define TMIX_COLSTATS_ROWS 32   /* rows per partial of col_stats_out */
int tmix_gemm_fp8(const tmix_gemm_desc* d, const uint8_t* scale_a, const uint8_t* scale_w, void* stream);
/* attn2 of a BasicTransformerBlock in ONE launch: q = to_q(LayerNorm(h)) and softmax(q K^T * scale) V against the cached prompt keys -- the patched
 * cross-attention forward of utils_custom.py:56-106 / utils_lora.py:65-69,101-111 (per-row concept weights = per-slice weight sets of d) without the q
 * round trip through memory and without the second launch.  d describes the projection exactly as for tmix_gemm_bf16 (A = hidden state rows, W = to_q
 * weight(s), bias, folded LayerNorm via ln_stats / ln_colsum, batch / strideW / w_period; no residual, activation, transposed region or statistics;
 * d->C is not written and may be NULL); N = heads * 64 must be a multiple of 320 (five heads per tile), K %% 64 == 0.  K: cached keys
 * [images][Skv][ldk] (head h at columns h*64..), Vt: cached V^T [images][N][ldvt] with ldvt == 80 (zero behind Skv), Skv <= 80; image of row r of
 * slice b = b * (M / rows_per_image) + r / rows_per_image (rows_per_image %% 64 == 0: the latent's token count).  O [batch * M][ldo] bf16 receives the
 * attention output (head h at columns h*64..), the A operand of attn2.to_out. */
int tmix_gemm_q_cross_attn(const tmix_gemm_desc* d, const void* K, int64_t ldk, int64_t strideK, const void* Vt, int64_t ldvt, int64_t strideVt,
                           void* O, int64_t ldo, int rows_per_image, int Skv, float scale, void* stream);
/* Row quantiser for tmix_gemm_fp8: X bf16 [rows][ld] -> Q e4m3 [rows][ldq] and scale_e8m0[r] = the smallest exponent that brings
 * max|X[r]| under 448 (K %% 8 == 0, K <= 8192).  Used on activations before each fp8 GEMM and once on the weights. */
int tmix_quantize_fp8_rows(const void* X, int64_t ld, void* Q, int64_t ldq, uint8_t* scale_e8m0, int64_t rows, int K, void* stream);
/* workgroup tile of a TMIX_TILE_* id (bm x bn), and the number of row-statistics partials a GEMM of width N writes with it */
int tmix_gemm_tile_shape(int tile_cfg, int* bm, int* bn);
int tmix_gemm_stats_parts(int N, int tile_cfg);

/* ---------------------------------------------------------------------------------------------
 * 3x3 convolution, NHWC bf16, implicit GEMM on MFMA (pad 1).
 * Replaces diffusers ResnetBlock2D.conv1/conv2, Downsample2D.conv (mode 1) and Upsample2D
 * (nearest x2 + conv, mode 2) inside the same UNet call sites.  Cin % 64 == 0.
 */
/* TMIX_CONV_T3: temporal convolution, kernel (3,1,1) with padding (1,0,0) over the FIRST spatial axis -- X is
 * [clips][frames][h*w][Cin] and Wt [Cout][3][Cin] (diffusers TemporalConvLayer's Conv3d of the I2VGen-XL UNet, config #5). */
/* TMIX_CONV_S2A: stride 2 with the zero padding on the right / bottom edge only (diffusers Downsample2D(padding=0) of the VAE ENCODER:
 * F.pad(x, (0,1,0,1)) then a stride-2 conv), out[y][x] = sum w[ky][kx] in[2y+ky][2x+kx]. */
enum { TMIX_CONV_S1 = 0, TMIX_CONV_S2 = 1, TMIX_CONV_UP2 = 2, TMIX_CONV_T3 = 3, TMIX_CONV_S2A = 4 };
typedef struct {
    const void* X;   /* bf16 [B][H][W][Cin]                                   */
    const void* Wt;  /* bf16 [Cout][3][3][Cin]  ([Cout][3][Cin] for T3)       */
    void*       Y;   /* bf16 [B][Ho][Wo][Cout]  Ho = H (S1), H/2 (S2), 2H (UP2) */
    const float* bias;           /* fp32 [Cout] or NULL                       */
    const float* batch_bias;     /* fp32 [B][Cout] (time embedding) or NULL   */
    const void* residual;        /* bf16 like Y or NULL                       */
    int32_t B, H, W, Cin, Cout, mode;
    int32_t tile_cfg;            /* TMIX_TILE_*                                */
    int32_t batch_bias_images;   /* consecutive images that share one batch_bias row (0/1: one row per image; the video
                                    UNet folds frames into B and has one time embedding per clip: = frames) */
    float* col_stats_out;        /* NULL or fp32 [B*Ho*Wo / 32][2][Cout]: as in tmix_gemm_desc (B*Ho*Wo %% 32 == 0, Cout %% 8 == 0) */
    /* --- the 1x1 conv_shortcut of a ResnetBlock2D in the same launch (diffusers ResnetBlock2D.forward: conv2(h) + conv_shortcut(x); in the
     * up-blocks x = cat[hidden, skip]): S1 / S2 = NULL or bf16 NHWC [B][H][W][S1_channels] / [..][S2_channels] (TMIX_CONV_S1 geometry only,
     * channels %% 64 == 0, S2 only with S1), and Wt then holds rows [Cout][9*Cin + S1_channels + S2_channels] = [conv2 taps | shortcut
     * weights over S1's channels | over S2's]; bias = conv2.bias + conv_shortcut.bias.  The K loop walks the two tensors' channels at the
     * output pixel behind the nine taps: one accumulator, no shortcut GEMM, no residual round trip, and the concatenation is never written. */
    const void* S1; const void* S2;
    int32_t S1_channels, S2_channels;
} tmix_conv_desc;
int tmix_conv3x3_nhwc(const tmix_conv_desc* d, void* stream);
/* the same convolution (diffusers ResnetBlock2D.conv1 / conv2 behind fusion_sampling.py:340) on OCP e4m3 operands: X = e4m3 bytes [B,H,W,Cin] with one E8M0
 * scale per (pixel, 32 channels) in the ROW-major form scale_x [B*H*W][Cin/32] -- what tmix_groupnorm_nhwc_pre_f8 writes -- Wt = e4m3 bytes [Cout][taps*Cin] with one
 * E8M0 scale per output channel (tmix_quantize_fp8_rows over the weight rows).  Cin %% 128 == 0 (a K-tile is 128 channels of one tap), no shortcut taps;
 * Y, bias, batch_bias, residual, col_stats_out as in tmix_conv3x3_nhwc.  tile_cfg: 12 or 20 (128x160 without / with two loader waves). */
int tmix_conv3x3_nhwc_fp8(const tmix_conv_desc* d, const uint8_t* scale_x, const uint8_t* scale_w, void* stream);

/* conv_in: fp32 NCHW latent [B,4,H,W] -> bf16 NHWC [B,H,W,Cout] (Cout % 32 == 0); weights fp32 OHWI. */
int tmix_conv_in(const float* x_nchw, const float* w_ohwi, const float* bias, void* y_nhwc,
                 int B, int Cin, int H, int W, int Cout, void* stream);
/* same with a per-pixel 4x4 linear map applied to the latent first (host pointers: pre_w[16] row-major, pre_b[4]):
 * the VAE's 1/scaling_factor and post_quant_conv (AutoencoderKL.decode), exact at the zero-padded border. */
int tmix_conv_in_pre(const float* x_nchw, const float* w_ohwi, const float* bias, void* y_nhwc,
                     int B, int Cin, int H, int W, int Cout, const float* pre_w, const float* pre_b, void* stream);
/* conv_out: bf16 NHWC [B,H,W,Cin] -> fp32 NCHW [B,Cout<=8,H,W]; weights bf16 OHWI. */
int tmix_conv_out(const void* x_nhwc, const void* w_ohwi, const float* bias, float* y_nchw,
                  int B, int Cin, int H, int W, int Cout, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Flash attention forward, head dim 64, bf16 in/out, fp32 softmax (never materialises S x S).
 * Replaces the explicit einsum/softmax/einsum of utils_custom.py:93-103 and utils_lora.py:101-111
 * and xformers' memory-efficient attention for attn1 (fusion_sampling.py:120,133,210).
 *   Q  [B][Sq][ldq]   head h at columns h*64..h*64+63
 *   K  [B][Skv][ldk]  same column convention
 *   Vt [B][H*64][ldvt] V TRANSPOSED (row = h*64+d, column = key); ldvt >= Skv rounded up to 8 and
 *                      the padding columns must hold finite values
 *   O  [B][Sq][ldo]
 */
int tmix_attn_fwd(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                  const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                  int B, int H, int Sq, int Skv, float scale, void* stream);
/* the same attention with the output as the block-scaled A operand of the out-projection behind it (utils_lora.py:116-119 / utils_custom.py:104-106 on
 * e4m3 operands, tmix_gemm_fp8 + TMIX_F8_A_BLOCK_SCALES): O8 [B*Sq][ldo8] OCP e4m3 bytes (head h at columns h*64..), scales [H*2][ldScale >= B*Sq] one
 * E8M0 byte per (row, 32 columns), k-block major -- bit for bit what an MX quantiser makes of the bf16 tensor tmix_attn_fwd writes; no bf16 output. */
int tmix_attn_fwd_f8(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                     const void* Vt, int64_t ldvt, int64_t strideVt, void* O8, int64_t ldo8, void* scales, int64_t ldScale,
                     int B, int H, int Sq, int Skv, float scale, void* stream);
/* The same two launches with a caller-owned workspace for the KEY-SPLIT TAIL.  B * H * ceil(Sq / 128) work items run on 512 workgroup slots; where a last,
 * partly filled round remains (the reference's attn1 at SDXL 1024^2, utils_lora.py:101-111 at B = 4: 640 items at S = 1024, 1280 at S = 4096) its items
 * are cut into 2 or 4 key ranges, one workgroup each, whose partial (o, maximum, row sum) meet in the workspace; the item's last arriver adds them in range
 * order (a fixed order: results do not depend on arrival) and stores.  tmix_attn_split_ws_bytes: bytes that enable the split for a shape (0: the shape does
 * not split; then, and with ws = NULL, the calls are exactly the two above).  The workspace is zero-filled ONCE by the caller (its first 4 KiB are ticket
 * counters every launch leaves at zero) and belongs to one stream at a time.  Results differ from the unsplit launch by fp32 rounding of the merge only. */
int64_t tmix_attn_split_ws_bytes(int B, int H, int Sq, int Skv);
int tmix_attn_fwd_ws(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                     const void* Vt, int64_t ldvt, int64_t strideVt, void* O, int64_t ldo, int64_t strideO,
                     int B, int H, int Sq, int Skv, float scale, void* ws, int64_t ws_bytes, void* stream);
int tmix_attn_fwd_f8_ws(const void* Q, int64_t ldq, int64_t strideQ, const void* K, int64_t ldk, int64_t strideK,
                        const void* Vt, int64_t ldvt, int64_t strideVt, void* O8, int64_t ldo8, void* scales, int64_t ldScale,
                        int B, int H, int Sq, int Skv, float scale, void* ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Normalisation / small ops (diffusers GroupNorm(32)+SiLU, LayerNorm, Timesteps, time MLPs).
 */
/* ws: fp32 workspace of >= tmix_groupnorm_ws_floats(B, C1+C2, groups) floats. Two inputs are
 * normalised as one tensor concatenated along C (X2 may be NULL, C2 = 0). */
int64_t tmix_groupnorm_ws_floats(int B, int C, int groups);
int tmix_groupnorm_ws_chunks(int64_t HW);   /* statistics workgroups per image (a function of the image size only) */
int tmix_groupnorm_nhwc(const void* X1, int C1, const void* X2, int C2, void* Y, const float* gamma,
                        const float* beta, float* ws, int B, int64_t HW, int groups, float eps, int silu,
                        void* stream);
/* kernels one tmix_groupnorm_nhwc call issues for an image of HW pixels and C channels: 3 (statistics, combine, apply) or 1 -- images whose slice of a few groups
 * (HW x 8..256 channels <= 64 K elements: the 336- / 84-pixel frames of video_gen/pipeline_i2vgen_xl.py:680-722's third and fourth levels) is small are normalised by one
 * workgroup per (image, group set) in a single launch.  A function of the image's shape only, never of the batch. */
int tmix_groupnorm_nhwc_launches(int64_t HW, int C, int groups);
/* The same normalisation with the statistics pass replaced by the column partials the PRODUCERS of the tensor left behind
 * (tmix_gemm_desc.col_stats_out / tmix_conv_desc.col_stats_out): cs1 = fp32 [B*HW / 32][2][cs1_channels] covers channels
 * [0, cs1_channels), cs2 (NULL with cs2_channels = 0) the cs2_channels behind them; cs1_channels + cs2_channels = C1 + C2.  The split of
 * the statistics need not be the split of X: the up-blocks normalise ONE concatenated tensor whose halves were written by two launches.
 * HW %% 32 == 0.  Two launches (combine partials -> scale / shift per channel, apply) instead of three, and X is read once instead of
 * twice.  Same ws as tmix_groupnorm_nhwc. */
int tmix_groupnorm_nhwc_pre(const void* X1, int C1, const void* X2, int C2, void* Y, const float* gamma,
                            const float* beta, float* ws, int B, int64_t HW, int groups, float eps, int silu,
                            const float* cs1, int cs1_channels, const float* cs2, int cs2_channels, void* stream);
/* tmix_groupnorm_nhwc_pre with the normalised (+ SiLU) tensor written as e4m3 bytes Y8 [B*HW][C] + scales [B*HW][C/32] (row-major MX blocks; C %% 32 == 0):
 * the input of tmix_conv3x3_nhwc_fp8; bit for bit what an MX quantiser makes of the bf16 tensor tmix_groupnorm_nhwc_pre writes. */
int tmix_groupnorm_nhwc_pre_f8(const void* X1, int C1, const void* X2, int C2, void* Y8, void* scales, const float* gamma,
                               const float* beta, float* ws, int B, int64_t HW, int groups, float eps, int silu,
                               const float* cs1, int cs1_channels, const float* cs2, int cs2_channels, void* stream);
int tmix_layernorm(const void* X, void* Y, const float* gamma, const float* beta, int64_t rows, int C,
                   float eps, void* stream);
/* hipMemsetAsync(ptr, 0, nbytes) on the stream (graph-capturable): zeroes the LayerNorm statistics accumulators */
int tmix_zero(void* ptr, int64_t nbytes, void* stream);
int tmix_concat_channels(const void* X1, int C1, const void* X2, int C2, void* Y, int64_t rows, void* stream);
/* sinusoidal embedding (flip_sin_to_cos=True, shift 0): out[i] = [cos(v_i f_j) | sin(v_i f_j)], j<dim/2 */
int tmix_timestep_embedding(const float* values, float* out, int count, int dim, void* stream);
/* row softmax: P[r][c] = softmax_c(scale * S[r][c]), S fp32 [rows][ld_s] -> P bf16 [rows][ld_p] (VAE mid-block attention,
 * single head of dim 512: scores come from tmix_gemm_bf16 with TMIX_EPI_F32OUT). */
int tmix_softmax_rows(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int cols, float scale, void* stream);
/* causal variant for the CLIP text encoders (transformers CLIPAttention with the causal mask, fusion_sampling.py:43-68):
 * S is a stack of [seq][cols] score blocks; row r sees columns <= r % seq, every other column (padding included) gets 0. */
int tmix_softmax_rows_causal(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int cols, float scale,
                             int seq, void* stream);
/* padded variant (CLIP vision tower of the video pipeline: 257 tokens in rows padded to a multiple of the GEMM K-tile):
 * columns >= valid get probability 0. */
int tmix_softmax_rows_masked(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int cols, int valid, float scale,
                             void* stream);
/* temporal self-attention over the frame axis (diffusers TransformerTemporalModel of the I2VGen-XL UNet, BASELINE config #5;
 * the reference drives it through video_gen/pipeline_i2vgen_xl.py:688-697): QKV bf16 [(clips*frames)][hw][ld] with columns
 * [0,C) = Q, [C,2C) = K, [2C,3C) = V, C = heads*64; every (clip, pixel, head) attends over its <= 16 frames;
 * O bf16 [(clips*frames)][hw][ldo], columns [0,C). */
int tmix_temporal_attn(const void* QKV, int64_t ld, void* O, int64_t ldo, int clips, int frames, int64_t hw, int heads,
                       float scale, void* stream);
/* LoRA in its low-rank form (utils_lora.py:65-79,113-119: batch row i + 1 gets  proj(x) + up_i(down_i(x)), model_lora.py:41-48, rank 4) without
 * merged per-concept weight copies: the attention projection runs once on shared weights [W | U | 0] over K + 64 input columns, and this
 * launch fills the 64 PAD columns behind every row of A with that row's down-projections: rows [b * rows_per_set, +rows_per_set) belong to
 * concept set sets[b] (0 = base, no delta); the P = 4 x (projections fused in the GEMM: 1, or 3 for q|k|v) values  A[m][0..K) . D[set * P + q][0..K)
 * go to columns K + set * P + q, every other pad column gets 0.  D is bf16 [nsets * P][K], nsets * P <= 64, P in {4, 12}.
 * Folded LayerNorm (dcolsum / dbias non-NULL; the GEMM then reads raw rows with ln_stats): the pad holds
 * (x - mean) D'^T + dbias / rstd  with D' = D * gamma as passed in D, dcolsum[i] = sum_k D'[i][k], dbias[i] = sum_k D[i][k] beta[k], and mean / rstd
 * of the row itself -- so that the GEMM's  rstd * (acc - mean * ln_colsum) + bias  (ln_colsum over the first K columns) adds exactly up(down(LN(x))). */
int tmix_lora_down(void* A, int64_t lda, int K, int64_t rows, const void* D, int P, int nsets, const float* dcolsum, const float* dbias,
                   float eps, const int* sets, int64_t rows_per_set, void* stream);
/* y = clamp(x*scale + shift, lo, hi) on fp32 (image post-processing (img/2+0.5).clamp(0,1), fusion_sampling.py:302) */
int tmix_affine_clamp(const float* x, float* y, int64_t n, float scale, float shift, float lo, float hi, void* stream);
/* out[M,N] = act_out( act_in(in[M,K]) * W[N,K]^T + bias ), fp32 activations, bf16 weights, M <= 256 (16 rows per launch).
 * act: 0 none, 1 SiLU.  add (nullable): fp32 [M,N] added before act_out. */
int tmix_linear_small(const float* in, const void* W, const float* bias, const float* add, float* out,
                      int M, int N, int K, int act_in, int act_out, void* stream);
/* The same for several weight matrices stacked along N that share the input (every ResnetBlock2D's time_emb_proj of one UNet
 * forward: 19 launches -> 1): sec_starts is a DEVICE array of nsec + 1 ascending column offsets (sec_starts[0] = 0,
 * sec_starts[nsec] = N); section s leaves as its own dense fp32 [M][width_s] matrix at out + sec_starts[s] * M. */
int tmix_linear_small_sections(const float* in, const void* W, const float* bias, float* out, int M, int N, int K,
                               int act_in, const int* sec_starts, int nsec, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sample-independent half of I2VGenXLUNet.forward (BASELINE config #5): the fps embedding, the context tokens and the image-latent
 * features that video_gen/pipeline_i2vgen_xl.py:604-639 prepares and :688-697 passes to the UNet on every step are constant over the
 * loop; tweediemix_amd/i2vgen.py conditioning() evaluates them once per video on these fp32 kernels (layers of 4 .. 64 channels).
 * 3x3 convolution, padding 1, stride 1 or 2: x fp32 NCHW [B,Cin,H,W], w fp32 OIHW [Cout,Cin,3,3] (the checkpoint's layout),
 * y fp32 NCHW [B,Cout,(H-1)/stride+1,(W-1)/stride+1]; silu != 0 applies SiLU to the result. */
int tmix_conv3x3_f32(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nchw, int B, int Cin, int H, int W, int Cout,
                     int stride, int silu, void* stream);
/* torch.nn.AdaptiveAvgPool2d((OH, OW)) over `planes` fp32 [H][W] planes: window i = [floor(i H / OH), ceil((i + 1) H / OH)). */
int tmix_adaptive_avgpool_f32(const float* x, float* y, int64_t planes, int H, int W, int OH, int OW, void* stream);
/* out[M,N] = act_out(act_in(in[M,K]) W[N,K]^T + bias), everything fp32 (act: 0 none, 1 SiLU), M <= 256. */
int tmix_linear_f32(const float* in, const float* W, const float* bias, float* out, int M, int N, int K, int act_in, int act_out, void* stream);
/* image_latents_temporal_encoder (diffusers I2VGenXLTransformerTemporalEncoder, width `channels` = 4): per (clip, pixel) over its <= 16 frames
 *   x1 = x + to_out(attention(LayerNorm(x)))  (two heads of 4, scale 1/2),   y = x1 + W2 gelu(W1 x1 + b1) + b2
 * x fp32 [clips*frames, 4, H, W] (the output of image_latents_proj_in), y fp32 [clips, 4, frames, H, W] (what the UNet concatenates to the
 * sample); weights fp32 in the checkpoint's [out, in] layout: wq / wk / wv [8,4], wo [4,8], w1 [16,4], w2 [4,16]. */
int tmix_i2v_temporal_encoder(const float* x, float* y, int clips, int frames, int channels, int64_t hw, const float* ln_gamma, const float* ln_beta,
                              const float* wq, const float* wk, const float* wv, const float* wo, const float* bo,
                              const float* w1, const float* b1, const float* w2, const float* b2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TMIX_H */
