import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 1280, 1280), (4096, 1280, 5120), (4096, 1280, 20480), (8192, 2560, 1280), (2048, 1280, 1280)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    res = []
    for cfg in (1, 2, 4, 7):
        d = ops.make_gemm_desc(a, w, out, tile_cfg=cfg)
        st = torch.cuda.current_stream().cuda_stream
        us = timeit(lambda: lib.tmix_gemm_bf16(C.byref(d), st))
        res.append(f"cfg{cfg}: {us:8.1f}us {2*M*N*K/us/1e6:6.0f}TF")
    print(f"{M}x{N}x{K}: " + " | ".join(res), flush=True)
