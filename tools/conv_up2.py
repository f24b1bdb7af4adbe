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
for (B, H, W, Cc, mode) in ((4, 32, 32, 640, 2), (4, 64, 64, 640, 2), (32, 32, 32, 640, 2), (32, 28, 48, 640, 2), (32, 28, 48, 640, 0), (32, 56, 96, 640, 0), (32, 28, 32, 640, 2), (32, 32, 48, 640, 2)):
    Ho, Wo = ops.conv_out_hw(H, W, mode)
    x = torch.randn(B, H, W, Cc, device="cuda").to(BF); w = (torch.randn(Cc, 3, 3, Cc, device="cuda") * (9 * Cc) ** -0.5).to(BF)
    out = torch.empty(B, Ho, Wo, Cc, device="cuda", dtype=BF); bias = torch.randn(Cc, device="cuda")
    row = []
    for cfg in (1, 2, 7):
        us = t(ops.make_conv_desc(x, w, out, bias, mode=mode, tile_cfg=cfg))
        row.append(f"c{cfg}:{us:7.1f}us/{2 * B * Ho * Wo * Cc * 9 * Cc / us / 1e6:4.0f}TF")
    print(f"conv B={B} {H}x{W} C={Cc} mode={mode}: " + " ".join(row), flush=True)
