"""per-shape throughput of tmix_gemm_bf16 / tmix_conv3x3_nhwc for every tile config (real SDXL 1024^2, B=4 shapes)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load()
BF = torch.bfloat16
dev = "cuda"

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

shapes = [  # (name, batch, M, N, K, geglu, trans)
    ("qkv1280", 4, 1024, 3840, 1280, False, True), ("proj1280", 1, 4096, 1280, 1280, False, False),
    ("ff1_1280", 1, 4096, 10240, 1280, True, False), ("ff2_1280", 1, 4096, 1280, 5120, False, False),
    ("qkv640", 4, 4096, 1920, 640, False, True), ("proj640", 1, 16384, 640, 640, False, False),
    ("ff1_640", 1, 16384, 5120, 640, True, False), ("ff2_640", 1, 16384, 640, 2560, False, False),
    ("sc320_640", 1, 16384, 640, 320, False, False),
]
only = sys.argv[1:] 
for name, b, M, N, K, geglu, trans in shapes:
    if only and name not in only: continue
    a = torch.randn(b, M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    res = []
    for cfg in (1, 2, 4, 7):
        if trans:
            Cq = N // 3
            out = torch.empty(b, M, 2 * Cq, device=dev, dtype=BF); vt = torch.zeros(b, Cq, M, device=dev, dtype=BF)
            d = ops.make_gemm_desc(a, w, out, out_t=vt, n_trans_begin=2 * Cq, tile_cfg=cfg)
        elif geglu:
            out = torch.empty(b, M, N // 2, device=dev, dtype=BF)
            d = ops.make_gemm_desc(a, w, out, geglu=True, tile_cfg=cfg)
        else:
            out = torch.empty(b, M, N, device=dev, dtype=BF)
            d = ops.make_gemm_desc(a, w, out, residual=out, tile_cfg=cfg)
        st = torch.cuda.current_stream().cuda_stream
        us = timeit(lambda: lib.tmix_gemm_bf16(C.byref(d), st))
        res.append(f"cfg{cfg}: {us:7.1f}us {2*b*M*N*K/us/1e6:6.0f}TF")
    print(f"{name:10s} b={b} M={M} N={N} K={K}: " + " | ".join(res), flush=True)

convs = [("c1280", 4, 32, 32, 1280, 1280, 0), ("c2560_1280", 4, 32, 32, 2560, 1280, 0), ("c640", 4, 64, 64, 640, 640, 0),
         ("c1920_640", 4, 64, 64, 1920, 640, 0), ("c320", 4, 128, 128, 320, 320, 0), ("c960_320", 4, 128, 128, 960, 320, 0),
         ("up1280", 4, 32, 32, 1280, 1280, 2), ("down320", 4, 128, 128, 320, 320, 1)]
for name, B, H, W, Ci, Co, mode in convs:
    if only and name not in only: continue
    x = torch.randn(B, H, W, Ci, device=dev).to(BF)
    w = (torch.randn(Co, 3, 3, Ci, device=dev) * (9 * Ci) ** -0.5).to(BF)
    Ho, Wo = ops.conv_out_hw(H, W, mode)
    out = torch.empty(B, Ho, Wo, Co, device=dev, dtype=BF)
    res = []
    for cfg in (1, 2, 4, 7):
        d = ops.make_conv_desc(x, w, out, mode=mode, tile_cfg=cfg)
        st = torch.cuda.current_stream().cuda_stream
        us = timeit(lambda: lib.tmix_conv3x3_nhwc(C.byref(d), st), n=10)
        res.append(f"cfg{cfg}: {us:7.1f}us {2*B*Ho*Wo*Co*9*Ci/us/1e6:6.0f}TF")
    print(f"{name:10s} {B}x{H}x{W} {Ci}->{Co} mode{mode}: " + " | ".join(res), flush=True)
