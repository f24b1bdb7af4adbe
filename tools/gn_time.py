import os, sys, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops
for (B, HW, C) in ((2, 16384, 320), (2, 4096, 640), (2, 1024, 1280), (1, 1024, 1280), (2, 1024, 2560), (2, 4096, 1920), (2, 16384, 960), (4, 4096, 640), (1, 16384, 512), (32, 5376, 320), (32, 1344, 640), (2, 86016, 320)):
    x = torch.randn(B, HW, C, device="cuda").to(torch.bfloat16); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    ws = ops.groupnorm_ws(B, C, 32, "cuda"); out = torch.empty_like(x)
    f = lambda: ops.groupnorm(x, g, b, 32, 1e-5, True, out=out, ws=ws)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"groupnorm B={B} HW={HW} C={C}: {us:6.1f} us  {3 * x.numel() * 2 / us / 1e6:5.2f} TB/s (3 passes)")
