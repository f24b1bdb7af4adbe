// gemm_inst_4.hip -- instantiations of gemm_conv_kernel (gemm_kernel.h) for one group of tilings
#include "gemm_kernel.h"

namespace tmix_gemm {

int launch_group4(int cfg, int conv, int f8, Params& p, int batch, hipStream_t st) {
    // phase-offset mainloop: bf16 (PH = 1) and fp8 (PH = 2), plain GEMM only
    if (conv) return -999;
    // f8: 0 = bf16, 1 = e4m3 with per-row scales, 2 = e4m3 with MX block scales on A
    if (cfg == 16) return f8 == 2 ? launch_cs<256, 256, 2, 4, 4, 0, 0, 3, 1, 0>(p, batch, st) : f8 ? launch_cs<256, 256, 2, 4, 4, 0, 0, 2, 1, 0>(p, batch, st) : launch_cs<256, 256, 2, 4, 4, 0, 0, 1, 1, 0>(p, batch, st);
    if (cfg == 22 && !f8) return launch_cs<256, 320, 4, 2, 4, 0, 0, 1, 1, 0>(p, batch, st);
    if (cfg == 17) return f8 == 2 ? launch_cs<256, 128, 4, 2, 4, 0, 0, 3, 1, 0>(p, batch, st) : f8 ? launch_cs<256, 128, 4, 2, 4, 0, 0, 2, 1, 0>(p, batch, st) : launch_cs<256, 128, 4, 2, 4, 0, 0, 1, 1, 0>(p, batch, st);
    return -999;
}

}  // namespace tmix_gemm
