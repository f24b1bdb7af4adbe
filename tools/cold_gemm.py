"""how much of the in-sequence GEMM time is cold-operand latency?  time one launch of a GEMM shape after (a) flushing the
caches with 1 GiB of junk traffic, (b) flush + a read pass over W (and A) that leaves them in the memory-side cache,
(c) back-to-back repeats (everything hot)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
junk = torch.empty(1 << 29, device="cuda", dtype=torch.int16)          # 1 GiB
def once(d, pre):
    ts = []
    for _ in range(5):
        junk.add_(1)                                                    # evict L2 + MALL
        for t in pre: t.float().sum()                                   # optional warm read
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.tmix_gemm_bf16(C.byref(d), st); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)
def hot(d):
    for _ in range(3): lib.tmix_gemm_bf16(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): lib.tmix_gemm_bf16(C.byref(d), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 100
for (M, N, K, cfg) in ((2048, 1280, 1280, 3), (2048, 1280, 5120, 3), (2048, 3840, 1280, 2), (2048, 10240, 1280, 7), (4096, 1280, 1280, 7), (4096, 1280, 5120, 2), (8192, 640, 640, 7), (8192, 5120, 640, 7)):
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    d = ops.make_gemm_desc(a, w, out, tile_cfg=cfg)
    print(f"{M}x{N}x{K} cfg{cfg}: cold {once(d, []):6.1f} us | W warm {once(d, [w]):6.1f} | A warm {once(d, [a]):6.1f} | A+W warm {once(d, [a, w]):6.1f} | hot loop {hot(d):6.1f}", flush=True)
