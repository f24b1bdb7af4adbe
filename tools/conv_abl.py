"""time the SDXL-level convolutions per tiling, hot (the launch repeats on the same operands); run once per library build, e.g.
TMIX_LIB=tools/ab/abl2/libtmix_hip.so python tools/conv_abl.py   (ablation builds: 2 = no MFMAs, 4 = no LDS-DMA in the loop of the non-loader tilings, 6 = neither)"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
def t(d, reps=10):
    for _ in range(2): lib.tmix_conv3x3_nhwc(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.tmix_conv3x3_nhwc(C.byref(d), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("library:", os.environ.get("TMIX_LIB", "shipped"))
for (B, H, W, Ci, Co) in ((4, 32, 32, 1280, 1280), (4, 32, 32, 2560, 1280), (4, 64, 64, 640, 640), (4, 64, 64, 1280, 640), (4, 128, 128, 320, 320)):
    x = torch.randn(B, H, W, Ci, device="cuda").to(BF); w = (torch.randn(Co, 3, 3, Ci, device="cuda") * (9 * Ci) ** -0.5).to(BF)
    out = torch.empty(B, H, W, Co, device="cuda", dtype=BF); bias = torch.randn(Co, device="cuda")
    row = []
    for cfg in (12, 20, 14, 7):
        us = t(ops.make_conv_desc(x, w, out, bias, mode=0, tile_cfg=cfg))
        row.append(f"c{cfg}:{us:7.1f}us/{2 * B * H * W * Co * 9 * Ci / us / 1e6:4.0f}TF ({us / (9 * Ci // 64) * 1e3:5.0f} ns/K-tile)")
    if Ci % 128 == 0 and hasattr(lib, "tmix_conv3x3_nhwc_fp8"):          # the same convolution on e4m3 operands (random bytes / scales: timing only)
        x8 = torch.randint(0, 120, (B, H, W, Ci), device="cuda", dtype=torch.uint8); sx = torch.full((B * H * W, Ci // 32), 127, device="cuda", dtype=torch.uint8)
        w8 = torch.randint(0, 120, (Co, 3, 3, Ci), device="cuda", dtype=torch.uint8); sw = torch.full((Co,), 120, device="cuda", dtype=torch.uint8)
        for cfg in (12, 20):
            d = ops.make_conv_desc(x8, w8, out, bias, mode=0, tile_cfg=cfg, _fp8=True)
            for _ in range(2): lib.tmix_conv3x3_nhwc_fp8(C.byref(d), sx.data_ptr(), sw.data_ptr(), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): lib.tmix_conv3x3_nhwc_fp8(C.byref(d), sx.data_ptr(), sw.data_ptr(), st)
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            row.append(f"f8/c{cfg}:{us:7.1f}us/{2 * B * H * W * Co * 9 * Ci / us / 1e6:4.0f}TF")
    print(f"conv B={B} {H}x{W} {Ci}->{Co}: " + " ".join(row), flush=True)
