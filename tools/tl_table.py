"""print a gemm_lab `tl` log (sections '===== <variant>') as one line per (shape, tiling) with the variants side by side"""
import re, sys
cur=None; shape=None; rows={}
for ln in open(sys.argv[1]):
    if ln.startswith('====='): cur=ln.split()[1]; continue
    m=re.match(r'gemm (\S+) b1 (\S+)',ln)
    if m: shape=m.group(1); continue
    m=re.match(r'\s+cfg\s+(\d+): hot\s+([\d.]+) us.*cold\s+([\d.]+) us',ln)
    if m: cfg=m.group(1); hot=m.group(2); cold=m.group(3); continue
    m=re.search(r'span\s+([\d.]+) us.*prologue\s+([\d.]+)\s+loop\s+([\d.]+)\s+epilogue\s+([\d.]+)',ln)
    if m: rows.setdefault((shape,cfg),{})[cur]=(hot,cold,m.group(2),m.group(3),m.group(4))
for k,v in rows.items():
    print(f"{k[0]:>18s} cfg{k[1]:>3s}:", end='')
    for var,(hot,cold,pro,loop,epi) in v.items():
        print(f" {var:5s} hot {hot:>6s} cold {cold:>6s} pro {pro:>4s} loop {loop:>6s} epi {epi:>5s} |", end='')
    print()
