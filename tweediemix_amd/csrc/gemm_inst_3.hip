// gemm_inst_3.hip -- instantiations of gemm_conv_kernel (gemm_kernel.h) for one group of tilings
#include "gemm_kernel.h"

namespace tmix_gemm {

int launch_group3(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st) {
    if (cfg == 14) return conv ? launch_cs<256, 320, 4, 2, 2, 1>(p, batch, st) : launch_cs<256, 320, 4, 2, 2, 0>(p, batch, st);
    if (cfg == 15) return conv ? launch_cs<32, 160, 1, 5, 4, 1>(p, batch, st) : launch_cs<32, 160, 1, 5, 4, 0>(p, batch, st);
    // loader-wave variants (plain GEMM only): one / two extra waves issue every LDS-DMA instruction of the K loop
    if (!conv && cfg == 19) return launch_cs<128, 160, 4, 1, 3, 0, 1>(p, batch, st);
    if (cfg == 20) return conv ? launch_cs<128, 160, 4, 1, 4, 1, 2>(p, batch, st) : launch_cs<128, 160, 4, 1, 4, 0, 2>(p, batch, st);
    if (!conv && cfg == 21) return launch_cs<128, 160, 4, 1, 4, 0, 4>(p, batch, st);     // one loader per SIMD
    return -999;
}

}  // namespace tmix_gemm
