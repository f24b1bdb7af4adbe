#!/usr/bin/env python3
"""variant of a tile table in which every convolution the halo-patch kernel (tiling 26, gemm_convh.hip) can run -- stride 1 (with or without shortcut taps), W % 32 == 0,
H % 4 == 0, Cout % 160 == 0 -- asks for it:   python tools/table_with_halo.py in.json out.json [--only "4,128,128"] [--small-tiles]
(--only: B,H,W prefixes to restrict to; --small-tiles: only entries that run a 128-row-or-smaller tile today -- 1, 3, 7, 12, 13, 15, 20 -- i.e. keep 256 x 320 / 256 x 256 / 256 x 128
where the table chose them: hot, 256 x 320 beats the halo kernel on the 128 x 128 level and on co-batched M >= 16384, tools/convh_bench.py)"""
import ast, json, sys
src, dst = sys.argv[1], sys.argv[2]
only = None
if "--only" in sys.argv:
    only = [tuple(int(v) for v in s.split(",")) for s in sys.argv[sys.argv.index("--only") + 1].split(";")]
t = json.load(open(src))
n = 0
for k in list(t):
    kk = k[len("shared|"):] if k.startswith("shared|") else k
    tup = ast.literal_eval(kk)
    if tup[0] != "conv" or len(tup) not in (7, 8) or (len(tup) == 8 and "--no-shortcut" in sys.argv):     # (8: + the shortcut taps' channel count)
        continue
    _c, B, H, W, Cin, Cout, mode = tup[:7]
    if "--small-tiles" in sys.argv and t[k] not in (1, 3, 7, 12, 13, 15, 20):
        continue
    if k.startswith("shared|"):          # a chain that shares the chip with a sibling (video: two clips): the halo kernel takes a CU's whole LDS (137 KB) -- 336 vs 273 us on the 56 x 96 frames
        continue
    if mode == 0 and W % 32 == 0 and H % 4 == 0 and Cout % 160 == 0 and Cin % 64 == 0 and (only is None or any(tup[1:1 + len(o)] == o for o in only)):
        t[k] = 26
        n += 1
json.dump(dict(sorted(t.items())), open(dst, "w"), indent=0)
print(f"{n} convolution entries -> tiling 26; wrote {dst}")
