// gemm_inst_3.hip -- instantiations of gemm_conv_kernel (gemm_kernel.h) for one group of tilings
#include "gemm_kernel.h"

namespace tmix_gemm {

int launch_group3(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st) {
    if (cfg == 14) return conv ? launch_cfg<256, 320, 4, 2, 2, 1>(p, batch, st) : launch_cfg<256, 320, 4, 2, 2, 0>(p, batch, st);
    if (cfg == 15) return conv ? launch_cfg<32, 160, 1, 5, 4, 1>(p, batch, st) : launch_cfg<32, 160, 1, 5, 4, 0>(p, batch, st);
    // loader-wave variants (plain GEMM only)
    if (!conv && cfg == 8) return launch_cfg<128, 160, 4, 1, 2, 0, 1>(p, batch, st);
    if (!conv && cfg == 9) return launch_cfg<256, 128, 4, 2, 3, 0, 1>(p, batch, st);
    if (!conv && cfg == 10) return launch_cfg<128, 128, 2, 2, 2, 0, 1>(p, batch, st);
    if (!conv && cfg == 11) return launch_cfg<256, 256, 2, 4, 2, 0, 1>(p, batch, st);
    return -999;
}

}  // namespace tmix_gemm
