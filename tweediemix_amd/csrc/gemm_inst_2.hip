// gemm_inst_2.hip -- instantiations of gemm_conv_kernel (gemm_kernel.h) for one group of tilings
#include "gemm_kernel.h"

namespace tmix_gemm {

int launch_group2(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st) {
    if (cfg == 7) return conv ? launch_cs<128, 160, 4, 1, 2, 1>(p, batch, st) : launch_cs<128, 160, 4, 1, 2, 0>(p, batch, st);
    if (cfg == 12) return conv ? launch_cs<128, 160, 4, 1, 4, 1>(p, batch, st) : launch_cs<128, 160, 4, 1, 4, 0>(p, batch, st);
#ifdef TMIX_TILE13_NS2     // dev A/B builds: tiling 13 with a two-deep ring (57 KB of LDS: TWO workgroups per CU, 512 co-resident 64 x 160 tiles for 4096 x 1280)
    if (cfg == 13) return conv ? launch_cs<64, 160, 1, 5, 4, 1>(p, batch, st) : launch_cs<64, 160, 1, 5, 2, 0>(p, batch, st);
#else
    if (cfg == 13) return conv ? launch_cs<64, 160, 1, 5, 4, 1>(p, batch, st) : launch_cs<64, 160, 1, 5, 4, 0>(p, batch, st);
#endif
    if (!conv && cfg == 18) return launch_cs<128, 160, 4, 1, 4, 0, 0, 0, 2>(p, batch, st);     // tiling 12 + in-workgroup split-K
    return -999;
}

}  // namespace tmix_gemm
