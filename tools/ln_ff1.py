import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
def t(d, reps=30):
    for _ in range(5): lib.tmix_gemm_bf16(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.tmix_gemm_bf16(C.byref(d), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = []
for (M, Cc) in ((4096, 1280), (16384, 640)):
    a = torch.randn(M, Cc, device="cuda").to(BF)
    w8 = (torch.randn(8 * Cc, Cc, device="cuda") * Cc ** -0.5).to(BF)
    b8 = torch.randn(8 * Cc, device="cuda")
    f = torch.empty(M, 4 * Cc, device="cuda", dtype=BF)
    stats = torch.rand((Cc + 127) // 128, M, 2, device="cuda") * 100 + 100
    cs = torch.randn(8 * Cc, device="cuda")
    for cfg in (1, 2, 7):
        g0 = min(t(ops.make_gemm_desc(a, w8, f, bias=b8, geglu=True, tile_cfg=cfg)) for _ in range(3))
        g1 = min(t(ops.make_gemm_desc(a, w8, f, bias=b8, geglu=True, ln_stats=stats, ln_colsum=cs, tile_cfg=cfg)) for _ in range(3))
        out.append(f"M={M} cfg{cfg}: {g0:.1f}->{g1:.1f} (+{g1-g0:.1f})")
print(" | ".join(out))
