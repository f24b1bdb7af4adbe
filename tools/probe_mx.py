import torch, sys
sys.path.insert(0, '.')
from tweediemix_amd import ops
M, N, K = 64, 64, 128
one = 0x38  # e4m3 1.0
a8 = torch.full((M, K), one, dtype=torch.uint8, device="cuda")
w8 = torch.zeros(N, K, dtype=torch.uint8, device="cuda")
for n in range(N):
    w8[n, 2 * n] = one          # picks K column 2n
sw = torch.full((N,), 127, dtype=torch.uint8, device="cuda")
for tile in (16, 17):
    sa = torch.full((K // 32, M), 127, dtype=torch.uint8, device="cuda")
    sa[0] = 127; sa[1] = 128; sa[2] = 129; sa[3] = 130      # block b scaled by 2^b
    out = ops.gemm_fp8(a8, sa, w8, sw, tile_cfg=tile, a_block_scales=True).float()
    print("tile", tile, "row0:", out[0].tolist())
    print("rows equal:", bool((out == out[0]).all()))
