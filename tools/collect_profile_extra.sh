#!/bin/bash
# run on the GPU box: tools/collect_profile_extra.sh <tag>  -- what tools/collect_profile.sh does not cover:
# (1) rocprofv3 --kernel-trace --stats + MfmaUtil of the VIDEO step (BASELINE config 5, tools/video_bench.py); (2) MfmaUtil of the fp8 plan
tag=${1:-r4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ${tag}_video -- python tools/video_bench.py --steps 10 > $out/${tag}_video_bench_line.json 2> $out/video.log
timeout 900 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $out -o ${tag}_video_MFMA -- python tools/video_bench.py --steps 2 --warmup 1 > $out/pmc_video_MFMA.log 2>&1
timeout 900 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $out -o ${tag}_fp8_MFMA -- python bench.py --dtype fp8 --steps 2 --warmup 1 --no-graphs --kind lora --no-trajectory --no-cpu-baseline --no-video > $out/pmc_fp8_MFMA.log 2>&1
python - $out $tag <<'PY'
import csv, sys, json, collections, re
out, tag = sys.argv[1], sys.argv[2]
for name in ("video", "fp8"):
    dur = {}
    try:
        for r in csv.DictReader(open(f"{out}/{tag}_{name}_MFMA_kernel_trace.csv")):
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f"{out}/{tag}_{name}_MFMA_counter_collection.csv")):
            if r["Counter_Name"] != "MfmaUtil": continue
            n = r["Kernel_Name"]
            m = re.search(r"gemm_conv_kernel<([^>]*)>", n)
            if m:
                a = [x.strip() for x in m.group(1).split(",")]
                key = ("conv" if a[5] == "1" else "gemm") + ("_fp8" if int(a[7]) >= 2 else "") + "<" + ",".join(a) + ">"
            else:
                key = "gemm<qattn:64x320,to_q+cross-attention>" if "gemm_qattn_kernel" in n else "gemm<w22:128x160,2x2>" if "gemm_w22_kernel" in n else "conv<halo:4x32px,160>" if "conv_halo_kernel" in n else "attn_fwd" if "attn_fwd" in n else "attn_small" if "attn_small" in n else "temporal_attn" if "temporal" in n else None
            if key: agg[key].append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 1.0)))
        cls = collections.defaultdict(list)
        for k, v in agg.items(): cls[k.split("<")[0]] += v
        wavg = lambda v: sum(u * d for u, d in v) / sum(d for _u, d in v)
        res = {"per_kernel_class_time_weighted_percent": {k: wavg(v) for k, v in cls.items()},
               "per_instance": {k: {"launches": len(v), "time_weighted_percent": wavg(v), "total_ms": sum(d for _u, d in v) / 1e6} for k, v in sorted(agg.items())},
               "note": "rocprofv3 --pmc MfmaUtil, eager pass (launches serialised by the profiler); averages weighted by kernel duration"}
        json.dump(res, open(f"{out}/{tag}_{name}_mfma_util.json", "w"), indent=1)
        print(name, json.dumps(res["per_kernel_class_time_weighted_percent"]))
    except Exception as e:
        print(name, "failed:", repr(e))
PY
rm -f $out/*_kernel_trace.csv $out/*counter_collection.csv $out/*agent_info.csv
ls $out
