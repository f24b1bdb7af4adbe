"""loader-wave tilings against their plain counterparts on this path's GEMM / conv shapes (isolated launches)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
def t(fn, d, reps=20):
    for _ in range(3): fn(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn(C.byref(d), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K, geglu) in ((4096, 1280, 1280, False), (2048, 1280, 1280, False), (4096, 10240, 1280, True), (4096, 1280, 5120, False),
                         (16384, 640, 640, False), (16384, 5120, 640, True), (8192, 8192, 8192, False)):
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=BF); bias = torch.randn(N, device="cuda")
    row = []
    for cfg in (7, 8, 2, 9, 1, 10, 4, 11):
        us = min(t(lib.tmix_gemm_bf16, ops.make_gemm_desc(a, w, out, bias=bias, geglu=geglu, tile_cfg=cfg)) for _ in range(2))
        row.append(f"c{cfg}:{us:6.1f}us/{2 * M * N * K / us / 1e6:4.0f}TF")
    print(f"gemm {M}x{N}x{K}{' geglu' if geglu else ''}: " + " ".join(row), flush=True)
for (B, H, Ci, Co) in ((4, 32, 1280, 1280), (4, 64, 640, 640), (4, 128, 320, 320), (4, 64, 1920, 640)):
    x = torch.randn(B, H, H, Ci, device="cuda").to(BF); w = (torch.randn(Co, 3, 3, Ci, device="cuda") * (9 * Ci) ** -0.5).to(BF)
    out = torch.empty(B, H, H, Co, device="cuda", dtype=BF); bias = torch.randn(Co, device="cuda")
    row = []
    for cfg in (7, 8, 2, 9, 1, 10):
        us = min(t(lib.tmix_conv3x3_nhwc, ops.make_conv_desc(x, w, out, bias, tile_cfg=cfg)) for _ in range(2))
        row.append(f"c{cfg}:{us:6.1f}us/{2 * B * H * H * Co * 9 * Ci / us / 1e6:4.0f}TF")
    print(f"conv {B}x{H}x{H} {Ci}->{Co}: " + " ".join(row), flush=True)
