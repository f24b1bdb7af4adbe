"""the SDXL-level stride-1 convolutions (1024^2, B = 4: latent 128 / 64 / 32) hot, tiling 20 (tap-major gather per tap, two loader waves) against tiling 26
(gemm_convh.hip: the halo patch of every 64-channel chunk resident in LDS); with the time-embedding row, residual and column statistics of the real launches.
  python tools/convh_bench.py [reps]"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def t(d):
    for _ in range(3): L.check(lib.tmix_conv3x3_nhwc(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.tmix_conv3x3_nhwc(C.byref(d), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, H, W, Ci, Co) in ((4, 128, 128, 320, 320), (4, 128, 128, 640, 320), (4, 128, 128, 960, 320), (4, 64, 64, 320, 640), (4, 64, 64, 640, 640), (4, 64, 64, 1280, 640), (4, 64, 64, 1920, 640),
                          (4, 32, 32, 640, 1280), (4, 32, 32, 1280, 1280), (4, 32, 32, 2560, 1280), (32, 32, 32, 1280, 1280), (32, 128, 128, 320, 320)):
    x = torch.randn(B, H, W, Ci, device="cuda").to(BF); w = (torch.randn(Co, 3, 3, Ci, device="cuda") * (9 * Ci) ** -0.5).to(BF)
    out = torch.empty(B, H, W, Co, device="cuda", dtype=BF); bias = torch.randn(Co, device="cuda"); temb = torch.randn(B, Co, device="cuda")
    cs = ops.colstats_buf(B * H * W, Co, "cuda")
    row = []
    for cfg in (14, 20, 26):
        us = t(ops.make_conv_desc(x, w, out, bias, batch_bias=temb, mode=0, tile_cfg=cfg, col_stats_out=cs))
        row.append(f"c{cfg}:{us:7.1f}us/{2 * B * H * W * Co * 9 * Ci / us / 1e6:4.0f}TF ({us * 256 / (B * H * W // 128 * (Co // 160)) / (9 * Ci // 64) * 1e3:5.0f} ns/K-tile/round)")
    print(f"conv B={B} {H}x{W} {Ci}->{Co}: " + " ".join(row), flush=True)
