import os, sys, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load()
for (B, Cin, H, W, Co) in ((32, 8, 56, 96, 320), (4, 4, 128, 128, 320), (1, 3, 512, 512, 128)):
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Co, 3, 3, Cin, device="cuda"); b = torch.randn(Co, device="cuda")
    y = torch.empty(B, H * W, Co, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.tmix_conv_in(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, Cin, H, W, Co, st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"conv_in B={B} Cin={Cin} {H}x{W} Cout={Co}: {us:.1f} us  {2*B*H*W*Co*9*Cin/us/1e6:.2f} TF")
