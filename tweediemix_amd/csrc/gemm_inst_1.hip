// gemm_inst_1.hip -- instantiations of gemm_conv_kernel (gemm_kernel.h) for one group of tilings
#include "gemm_kernel.h"

namespace tmix_gemm {

int launch_group1(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st) {
    if (cfg == 4) return conv ? launch_cs<256, 256, 2, 4, 2, 1>(p, batch, st) : launch_cs<256, 256, 2, 4, 2, 0>(p, batch, st);
    if (cfg == 5) return conv ? launch_cs<256, 128, 2, 2, 3, 1>(p, batch, st) : launch_cs<256, 128, 2, 2, 3, 0>(p, batch, st);
    return -999;
}

}  // namespace tmix_gemm
