// gemm_lab -- standalone (no Python, no torch) timing + correctness harness for tmix_gemm_bf16 through the C ABI.
//   lab [check|time] M,N,K[,batch[,flags]] ... -- cfgs=1,2,16  reps=30
// flags (letters): b bias, r residual, g GEGLU, s row_stats_out, t transposed tail (last third of N, batch must divide M),
//                  f fp8 operands (quantised once by tmix_quantize_fp8_rows, then tmix_gemm_fp8 is timed)
// "hot"  : the same operands every launch (everything L2 / Infinity-Cache resident after the first pass)
// "cold" : the launch cycles through enough operand sets to exceed the 256 MiB Infinity Cache (weights and activations cold)
// Dev tool only: product code never links this.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <math.h>
#include "../../include/tmix.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;      // uniform [-scale, scale)
        uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u);
        p[i] = (uint16_t)(u >> 16);
    }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
        p[i] = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
    }
}
// naive reference: one thread per output, fp32 accumulate in k order
__global__ void ref_gemm(const uint16_t* A, const uint16_t* W, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += __uint_as_float(((uint32_t)A[(size_t)m * K + k]) << 16) * __uint_as_float(((uint32_t)W[(size_t)n * K + k]) << 16);
    C[(size_t)m * N + n] = acc;
}

struct Shape { int M, N, K, batch; std::string flags; };

int main(int argc, char** argv) {
    std::vector<Shape> shapes; std::vector<int> cfgs; int reps = 30; bool check = false, cold = true, hot = true; float dscale = 1.0f; bool timeline = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "check") check = true;
        else if (a == "nocold") cold = false;
        else if (a == "nohot") hot = false;
        else if (a == "zero") dscale = 0.0f;          // zero-filled operands (DVFS: the chip clocks higher on them)
        else if (a == "tl") timeline = true;          // per-workgroup phase breakdown through tmix_prof_begin/end
        else if (a.rfind("cfgs=", 0) == 0) { char* s = argv[i] + 5; for (char* t = strtok(s, ","); t; t = strtok(nullptr, ",")) cfgs.push_back(atoi(t)); }
        else if (a.rfind("reps=", 0) == 0) reps = atoi(argv[i] + 5);
        else if (a == "--" || a == "time") {}
        else { Shape s{0, 0, 0, 1, ""}; char fl[32] = ""; int n = sscanf(argv[i], "%d,%d,%d,%d,%31s", &s.M, &s.N, &s.K, &s.batch, fl); if (n < 3) { fprintf(stderr, "bad shape %s\n", argv[i]); return 2; } if (n < 4) s.batch = 1; s.flags = fl; shapes.push_back(s); }
    }
    if (cfgs.empty()) cfgs = {1, 2, 7, 16};
    if (tmix_check_device() != 0) { fprintf(stderr, "device check failed: %s\n", tmix_last_error_string()); return 2; }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int rc_all = 0;
    for (const Shape& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, B = sh.batch;
        const bool fb = sh.flags.find('b') != std::string::npos, fr = sh.flags.find('r') != std::string::npos, fg = sh.flags.find('g') != std::string::npos;
        const bool fs = sh.flags.find('s') != std::string::npos, ft = sh.flags.find('t') != std::string::npos, f8 = sh.flags.find('f') != std::string::npos;
        const int Nout = fg ? N / 2 : N;
        const size_t szA = (size_t)B * M * K, szW = (size_t)N * K, szC = (size_t)B * M * Nout;
        const size_t set_bytes = 2 * (szA + szW + szC + (fr ? szC : 0));
        int nset = cold ? (int)std::min<size_t>(64, (600u << 20) / set_bytes + 1) : 1;
        if (nset < 1) nset = 1;
        std::vector<uint16_t*> As(nset), Ws(nset), Cs(nset), Rs(nset);
        for (int s = 0; s < nset; ++s) {
            CK(hipMalloc(&As[s], szA * 2)); CK(hipMalloc(&Ws[s], szW * 2)); CK(hipMalloc(&Cs[s], szC * 2)); CK(hipMalloc(&Rs[s], szC * 2));
            fill_bf16<<<1024, 256, 0, st>>>(As[s], szA, 17 + s, 1.0f * dscale);
            fill_bf16<<<1024, 256, 0, st>>>(Ws[s], szW, 91 + s, 1.0f / sqrtf((float)K) * 1.7f * dscale);
            fill_bf16<<<1024, 256, 0, st>>>(Rs[s], szC, 55 + s, 1.0f);
        }
        std::vector<uint8_t*> A8(nset, nullptr), W8(nset, nullptr), SA(nset, nullptr), SW(nset, nullptr);
        if (f8) for (int s = 0; s < nset; ++s) {
            CK(hipMalloc(&A8[s], szA)); CK(hipMalloc(&W8[s], szW)); CK(hipMalloc(&SA[s], (size_t)B * M)); CK(hipMalloc(&SW[s], N));
            if (tmix_quantize_fp8_rows(As[s], K, A8[s], K, SA[s], (int64_t)B * M, K, st) || tmix_quantize_fp8_rows(Ws[s], K, W8[s], K, SW[s], N, K, st)) {
                fprintf(stderr, "quantize: %s\n", tmix_last_error_string()); return 2; }
        }
        float* bias; CK(hipMalloc(&bias, N * 4)); fill_f32<<<64, 256, 0, st>>>(bias, N, 5, 0.5f);
        float* stats; CK(hipMalloc(&stats, (size_t)16 * B * M * 2 * 4));
        uint16_t* Ct = nullptr; const int ldct = (M + 7) / 8 * 8; const int ntb = ft ? (N / 3) * 2 : -1;
        if (ft) CK(hipMalloc(&Ct, (size_t)B * (N - ntb) * ldct * 2));
        CK(hipStreamSynchronize(st));
        auto mk = [&](int set, int cfg) {
            tmix_gemm_desc d; memset(&d, 0, sizeof d);
            d.A = f8 ? (void*)A8[set] : (void*)As[set]; d.lda = K; d.strideA = B > 1 ? (int64_t)M * K : 0;
            d.W = f8 ? (void*)W8[set] : (void*)Ws[set]; d.ldw = K; d.strideW = 0;
            d.C = Cs[set]; d.ldc = Nout; d.strideC = B > 1 ? (int64_t)M * Nout : 0;
            if (fb || fg) d.bias = bias;
            if (fr) { d.residual = Rs[set]; d.ldr = Nout; d.strideR = B > 1 ? (int64_t)M * Nout : 0; }
            d.n_trans_begin = ntb; if (ft) { d.Ct = Ct; d.ldct = ldct; d.strideCt = (int64_t)(N - ntb) * ldct; }
            d.M = M; d.N = N; d.K = K; d.batch = B; d.epilogue = fg ? TMIX_EPI_GEGLU : TMIX_EPI_NONE; d.tile_cfg = cfg;
            if (fs) { d.row_stats_out = stats; d.strideStatsOut = 2 * M; d.ldStatsOut = (int64_t)B * M; }
            return d;
        };
        auto run = [&](tmix_gemm_desc* d, int set) { return f8 ? tmix_gemm_fp8(d, SA[set], SW[set], st) : tmix_gemm_bf16(d, st); };
        std::vector<float> ref;
        if (check && !fg && !ft && !f8) {
            float* Cr; CK(hipMalloc(&Cr, (size_t)B * M * N * 4));
            for (int b = 0; b < B; ++b)
                ref_gemm<<<dim3((N + 255) / 256, M), 256, 0, st>>>(As[0] + (size_t)b * M * K, Ws[0], Cr + (size_t)b * M * N, M, N, K);
            ref.resize((size_t)B * M * N);
            CK(hipMemcpyAsync(ref.data(), Cr, ref.size() * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); CK(hipFree(Cr));
        }
        printf("gemm %dx%dx%d b%d %s  (%d operand sets, %.1f MB each)\n", M, N, K, B, sh.flags.c_str(), nset, set_bytes / 1048576.0);
        const double fl = 2.0 * B * M * N * K;
        for (int cfg : cfgs) {
            tmix_gemm_desc d0 = mk(0, cfg);
            int rc = run(&d0, 0);
            if (rc) { printf("  cfg %2d: rc %d (%s)\n", cfg, rc, tmix_last_error_string()); continue; }
            CK(hipStreamSynchronize(st));
            std::string verdict;
            if (!ref.empty()) {
                std::vector<uint16_t> out(szC), res(szC); std::vector<float> hb(N);
                CK(hipMemcpy(out.data(), Cs[0], szC * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(res.data(), Rs[0], szC * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hb.data(), bias, N * 4, hipMemcpyDeviceToHost));
                double maxe = 0, maxr = 0; size_t bad = 0;
                for (size_t i = 0; i < szC; ++i) {
                    float want = ref[i] + (fb ? hb[i % N] : 0.f) + (fr ? bf2f(res[i]) : 0.f);
                    const float got = bf2f(out[i]);
                    const double e = fabs((double)got - want), tol = 0.02 + 0.01 * fabs(want);
                    if (e > tol) ++bad;
                    maxe = std::max(maxe, e); maxr = std::max(maxr, (double)fabs(want));
                }
                char buf[96]; snprintf(buf, sizeof buf, "  check: max|err| %.4f (max|ref| %.1f) bad %zu%s", maxe, maxr, bad, bad ? "  <-- WRONG" : " ok");
                verdict = buf; if (bad) rc_all = 1;
            }
            double us_hot = 0, us_cold = 0;
            for (int mode = 0; mode < 2; ++mode) {
                if ((mode == 0 && !hot) || (mode == 1 && (!cold || nset < 2))) continue;
                std::vector<tmix_gemm_desc> ds; for (int s = 0; s < (mode ? nset : 1); ++s) ds.push_back(mk(s, cfg));
                for (int w = 0; w < 3; ++w) run(&ds[w % ds.size()], w % (int)ds.size());
                double best = 1e30;
                for (int round = 0; round < 3; ++round) {
                    CK(hipEventRecord(e0, st));
                    for (int r = 0; r < reps; ++r) run(&ds[r % ds.size()], r % (int)ds.size());
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, (double)ms * 1e3 / reps);
                }
                (mode ? us_cold : us_hot) = best;
            }
            printf("  cfg %2d: hot %7.1f us %6.0f TF | cold %7.1f us %6.0f TF%s\n", cfg, us_hot, us_hot ? fl / us_hot / 1e6 : 0, us_cold, us_cold ? fl / us_cold / 1e6 : 0, verdict.c_str());
            if (timeline) {
                const int NL = 6; uint64_t* slots; CK(hipMalloc(&slots, NL * 64));
                std::vector<uint64_t> init(NL * 8, 0); for (int i = 0; i < NL; ++i) init[i * 8] = ~0ull;
                CK(hipMemcpy(slots, init.data(), NL * 64, hipMemcpyHostToDevice));
                std::vector<tmix_gemm_desc> ds; for (int s = 0; s < nset; ++s) ds.push_back(mk(s, cfg));
                tmix_prof_begin(slots, NL, 1);
                for (int r = 0; r < NL; ++r) run(&ds[(r + 1) % ds.size()], (r + 1) % (int)ds.size());
                tmix_prof_end();
                CK(hipStreamSynchronize(st));
                std::vector<uint64_t> got(NL * 8); CK(hipMemcpy(got.data(), slots, NL * 64, hipMemcpyDeviceToHost)); CK(hipFree(slots));
                double span = 0, a1 = 0, a2 = 0, a3 = 0; int cnt = 0;
                for (int r = 1; r < NL; ++r) {       // skip the first
                    const uint64_t* g = &got[r * 8]; const double n = (double)g[5];
                    span += (g[1] - g[0]) * 0.01; a1 += g[2] / n * 0.01; a2 += g[3] / n * 0.01; a3 += g[4] / n * 0.01; ++cnt;
                }
                printf("          timeline (%s): kernel span %6.1f us | per workgroup: prologue %5.1f  loop %6.1f  epilogue %5.1f  (wg total %6.1f, %llu wgs)\n",
                       nset > 1 ? "cold" : "hot", span / cnt, a1 / cnt, (a2 - a1) / cnt, (a3 - a2) / cnt, a3 / cnt, (unsigned long long)got[13]);
            }
            fflush(stdout);
        }
        for (int s = 0; s < nset; ++s) { CK(hipFree(As[s])); CK(hipFree(Ws[s])); CK(hipFree(Cs[s])); CK(hipFree(Rs[s])); }
        CK(hipFree(bias)); CK(hipFree(stats)); if (Ct) CK(hipFree(Ct));
    }
    return rc_all;
}
