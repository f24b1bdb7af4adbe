#!/bin/bash
# round 4: LDS bank-conflict counters of the hot GEMM tilings (FF2 on 21 / 19, FF1 on 14, q/k/v on 16) and the conv (gemm_lab + conv_abl), one PMC pass each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r4zs; mkdir -p $out; rm -rf $out/*
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o lds -- tools/gemm_lab/lab time 4096,1280,5120,1,br 4096,10240,1280,1,g 4096,3840,1280,1,b cfgs=21,19,14,16 reps=3 nocold > $out/lab.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out -o lds2 -- tools/gemm_lab/lab time 4096,1280,5120,1,br 4096,10240,1280,1,g cfgs=21,14 reps=3 nocold > $out/lab2.log 2>&1
python - $out <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_conv_kernel" not in n: continue
        k = n[n.index("<"):n.index(">") + 1]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    print(f)
    for k, v in agg.items():
        print("  ", k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
PY
