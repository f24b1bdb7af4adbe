"""the video step's small-K projections (K = 320 / 640 at 86016 / 21504 rows: pure streaming launches) on every tiling, hot"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
def hot(d, n=20):
    for _ in range(3): lib.tmix_gemm_bf16(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.tmix_gemm_bf16(C.byref(d), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K) in ((86016, 320, 320), (86016, 960, 320), (21504, 640, 640), (21504, 1920, 640), (5376, 1280, 1280), (86016, 320, 1280)):
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(BF); out = torch.empty(M, N, device="cuda", dtype=BF)
    row = []
    for cfg in (1, 2, 4, 5, 7, 12, 14, 16, 17, 20, 21, 22, 23):
        try:
            d = ops.make_gemm_desc(a, w, out, bias=bias, residual=res, tile_cfg=cfg)
            row.append(f"{cfg}:{hot(d):6.1f}")
        except Exception as e:
            row.append(f"{cfg}: err")
    mb = (M * K + N * K + 2 * M * N) * 2 / 1e6
    print(f"M={M} N={N} K={K} ({mb:.0f} MB algorithmic = {mb / 4e3 * 1e3 / 1e3:.1f} us at 4 TB/s): " + " ".join(row), flush=True)
