// gemm_inst_5.hip -- instantiations of gemm_conv_kernel (gemm_kernel.h): e4m3 operands in the lock-step loops, 128 x 160 tiles
#include "gemm_kernel.h"

namespace tmix_gemm {

int launch_group5(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st) {
    if (f8 < 3) return -999;
    if (conv) {                  // tmix_conv3x3_nhwc_fp8: e4m3 input with row-major MX block scales (f8 = 4), 128-channel K-tiles
        if (cfg == 12) return launch_cs<128, 160, 4, 1, 4, 1, 0, 5>(p, batch, st);
        if (cfg == 20) return launch_cs<128, 160, 4, 1, 4, 1, 2, 5>(p, batch, st);
        return -999;
    }
    // f8: 3 = one E8M0 scale per A row, 4 = MX block scales on A (one more LDS-DMA piece per K-tile)
    if (cfg == 12) return f8 == 4 ? launch_cs<128, 160, 4, 1, 4, 0, 0, 5, 1, 0>(p, batch, st) : launch_cs<128, 160, 4, 1, 4, 0, 0, 4, 1, 0>(p, batch, st);
    if (cfg == 19) return f8 == 4 ? launch_cs<128, 160, 4, 1, 3, 0, 1, 5, 1, 0>(p, batch, st) : launch_cs<128, 160, 4, 1, 3, 0, 1, 4, 1, 0>(p, batch, st);
    if (cfg == 20) return f8 == 4 ? launch_cs<128, 160, 4, 1, 4, 0, 2, 5, 1, 0>(p, batch, st) : launch_cs<128, 160, 4, 1, 4, 0, 2, 4, 1, 0>(p, batch, st);
    if (cfg == 21) return f8 == 4 ? launch_cs<128, 160, 4, 1, 4, 0, 4, 5, 1, 0>(p, batch, st) : launch_cs<128, 160, 4, 1, 4, 0, 4, 4, 1, 0>(p, batch, st);
    return -999;
}

}  // namespace tmix_gemm
