#!/bin/bash
# round 6, VERDICT r5 item 6: (i) the N = K = 1280 out-projections as 512 co-resident 64 x 160 tiles (tiling 13 with a two-deep ring: two workgroups per CU) against tiling 21 / 23,
# hot and in the captured step; (ii) the one-launch attn2 on 64 x 256 four-head tiles (320 tiles, 40 KB K-tiles) against the five-head form, hot inside a graph
out=gpurun_out/r6v; mkdir -p $out
python - <<'PY' > $out/t13_table.json
import json
t = json.load(open("tweediemix_amd/tuned_gfx950.json"))
for k in ("('gemm', 1024, 1280, 1280, 4, 0, False, True, True, True, False)", "routed|('gemm', 1024, 1280, 1280, 4, 0, False, True, True, True, False)"):
    t[k] = 13
print(json.dumps(t))
PY
{
echo "== (i) hot: 4 x 1024 x 1280 x 1280 routed out-projection (residual, row statistics), tilings 21 / 23 / 13 (four-deep ring, one workgroup per CU) / 13 with a two-deep ring (two per CU)"
python - <<'PY'
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
def run(libpath, cfgs):
    os.environ["TMIX_LIB"] = libpath
    import importlib
    from tweediemix_amd import lib as L, ops
    lib = L.load(); BF = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
    a = torch.randn(4, 1024, 1280, device="cuda").to(BF); w = (torch.randn(4, 1280, 1280, device="cuda") * 1280 ** -0.5).to(BF)
    out = torch.empty(4, 1024, 1280, device="cuda", dtype=BF); res = torch.randn(4, 1024, 1280, device="cuda").to(BF); bias = torch.randn(1280, device="cuda")
    for cfg in cfgs:
        stt = torch.zeros(8, 4096, 2, device="cuda")
        d = ops.make_gemm_desc(a, w, out, bias=bias, residual=res, row_stats_out=stt, tile_cfg=cfg)
        for _ in range(3): L.check(lib.tmix_gemm_bf16(C.byref(d), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.tmix_gemm_bf16(C.byref(d), st)
        e1.record(); e1.synchronize()
        print(f"  {os.path.basename(os.path.dirname(libpath))} cfg {cfg}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
run(sys.argv[1] if len(sys.argv) > 1 else "tweediemix_amd/lib/libtmix_hip.so", (21, 23, 13))
PY
TMIX_LIB=tools/ab/t13ns2/libtmix_hip.so python - <<'PY'
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import lib as L, ops
lib = L.load(); BF = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
a = torch.randn(4, 1024, 1280, device="cuda").to(BF); w = (torch.randn(4, 1280, 1280, device="cuda") * 1280 ** -0.5).to(BF)
out = torch.empty(4, 1024, 1280, device="cuda", dtype=BF); res = torch.randn(4, 1024, 1280, device="cuda").to(BF); bias = torch.randn(1280, device="cuda")
stt = torch.zeros(8, 4096, 2, device="cuda")
d = ops.make_gemm_desc(a, w, out, bias=bias, residual=res, row_stats_out=stt, tile_cfg=13)
ref = ops.gemm(a, w, bias=bias, residual=res, tile_cfg=21)
for _ in range(3): L.check(lib.tmix_gemm_bf16(C.byref(d), st))
torch.cuda.synchronize(); print("  two-deep ring equals tiling 21:", bool(torch.equal(out, ref)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): lib.tmix_gemm_bf16(C.byref(d), st)
e1.record(); e1.synchronize()
print(f"  t13ns2 cfg 13 (two-deep ring, two workgroups per CU): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
PY
echo "== (i) in the captured step: shipped table / routed out-projections on tiling 13 (four-deep) / the same on the two-deep ring"
for r in 1 2; do
  echo -n "shipped: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "t13 four-deep: "; TMIX_TUNE_FILE=$out/t13_table.json python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "t13 two-deep, 2 WG/CU: "; TMIX_TUNE_FILE=$out/t13_table.json TMIX_LIB=tools/ab/t13ns2/libtmix_hip.so python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
done
echo "== (ii) one-launch attn2 hot inside a graph: five-head 64 x 320 tiles (shipped) / four-head 64 x 256 tiles"
python tools/qattn_bench.py 2>/dev/null | tail -4
TMIX_LIB=tools/ab/qattn_h4/libtmix_hip.so python tools/qattn_bench.py 2>/dev/null | tail -4
} 2>&1 | tee $out/item6.txt
