"""this path's GEMM shapes: tmix_gemm_bf16 (best tiling per shape, isolated hot loop) next to torch's bf16 matmul
(hipBLASLt / rocBLAS under the hood) as an outside reference point."""
import os, sys, json, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
rows = []
SHAPES = ((4096, 1280, 1280), (2048, 1280, 1280), (4096, 3840, 1280), (4096, 10240, 1280), (4096, 1280, 5120), (16384, 640, 640),
          (16384, 5120, 640), (16384, 640, 2560), (8192, 8192, 8192))
if "--cobatch" in sys.argv:      # 8 co-batched seeds (images/s path): the B = 32 fusion / start plans (M = 32768 at 32 x 32, 131072 at 64 x 64) and the B = 16 plain plan
    sys.argv.remove("--cobatch")
    SHAPES = tuple((M, N, K) for M in (32768, 16384) for (N, K) in ((1280, 1280), (3840, 1280), (10240, 1280), (1280, 5120))) + \
             tuple((M, N, K) for M in (131072, 65536) for (N, K) in ((640, 640), (1920, 640), (5120, 640), (640, 2560))) + ((8192, 8192, 8192),)
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF); bias = torch.randn(N, device="cuda")
    best = (1e9, 0)
    for cfg in (1, 2, 3, 4, 5, 7, 12, 13, 14, 16, 17, 18, 19, 20, 21, 22, 23):
        d = ops.make_gemm_desc(a, w, out, bias=bias, tile_cfg=cfg)
        best = min(best, (timeit(lambda: lib.tmix_gemm_bf16(C.byref(d), st)), cfg))
    tb = timeit(lambda: torch.matmul(a, w.t()))
    tl = timeit(lambda: torch.nn.functional.linear(a, w, bias.to(BF)))
    fl = 2 * M * N * K
    rows.append({'M': M, 'N': N, 'K': K, 'tmix_cfg': best[1], 'tmix_us': best[0], 'torch_matmul_us': tb, 'torch_linear_bias_us': tl})
    print(f"{M}x{N}x{K}: tmix cfg{best[1]:2d} {best[0]:7.1f}us {fl / best[0] / 1e6:5.0f}TF | torch.matmul {tb:7.1f}us {fl / tb / 1e6:5.0f}TF | F.linear+bias {tl:7.1f}us {fl / tl / 1e6:5.0f}TF", flush=True)
if len(sys.argv) > 1: json.dump({'what': 'isolated hot loops, bias epilogue, best tmix tiling vs torch (hipBLASLt)', 'rows': rows}, open(sys.argv[1], 'w'), indent=1)
