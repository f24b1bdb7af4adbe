// gemm_inst_0.hip -- instantiations of gemm_conv_kernel (gemm_kernel.h) for one group of tilings
#include "gemm_kernel.h"

namespace tmix_gemm {

int launch_group0(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st) {
    if (cfg == 1) return conv ? launch_cs<128, 128, 2, 2, 2, 1>(p, batch, st) : launch_cs<128, 128, 2, 2, 2, 0>(p, batch, st);
    if (cfg == 2) return conv ? launch_cs<256, 128, 4, 2, 3, 1>(p, batch, st) : launch_cs<256, 128, 4, 2, 3, 0>(p, batch, st);
    if (cfg == 3) return conv ? launch_cs<128, 128, 2, 2, 4, 1>(p, batch, st) : launch_cs<128, 128, 2, 2, 4, 0>(p, batch, st);
    return -999;
}

}  // namespace tmix_gemm
