import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(4096, 1280, 64), (4096, 1280, 128), (4096, 1280, 256), (4096, 1280, 640), (4096, 1280, 1280), (128, 128, 64), (128, 128, 1280), (2048, 1280, 64), (4096, 2560, 64)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    res = []
    for cfg in (1, 2):
        d = ops.make_gemm_desc(a, w, out, tile_cfg=cfg)
        us = timeit(lambda: lib.tmix_gemm_bf16(C.byref(d), st))
        res.append(f"cfg{cfg}: {us:6.1f}us")
    print(f"{M}x{N}x{K}: " + " | ".join(res), flush=True)
x = torch.zeros(1 << 20, device="cuda")
print("torch tiny kernel (fill_):", timeit(lambda: x[:64].fill_(1.0)), "us")
ln_x = torch.randn(4096, 1280, device="cuda").to(BF); g = torch.ones(1280, device="cuda"); b = torch.zeros(1280, device="cuda"); o = torch.empty_like(ln_x)
print("layernorm 4096x1280:", timeit(lambda: ops.layernorm(ln_x, g, b, out=o)), "us")
