#!/bin/bash
# run on the GPU box: tools/collect_profile.sh <tag>
# (1) rocprofv3 --kernel-trace --stats of the default bench command; (2) separate PMC passes for HBM traffic;
# (3) a PMC pass for rocprofiler's MfmaUtil (SQ_VALU_MFMA_BUSY_CYCLES summed / (GRBM_GUI_ACTIVE x SIMDs)) per kernel.
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
BARGS="--kind lora --no-trajectory --no-cpu-baseline --no-video ${BENCH_EXTRA:-}"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ${tag} -- python bench.py $BARGS > $out/${tag}_bench_line.json 2> $out/bench.log
# the fp8 plan (--dtype fp8: attn1 q/k/v, attn2 to_q and both FF projections on e4m3 operands) under the same kernel trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ${tag}_fp8 -- python bench.py --dtype fp8 $BARGS > $out/${tag}_fp8_bench_line.json 2> $out/bench_fp8.log
for pm in FETCH_SIZE WRITE_SIZE; do
  # (each counter pass under its own timeout: a pass that hangs costs its own evidence, not the others')
  timeout 240 rocprofv3 --pmc $pm --kernel-trace --output-format csv -d $out -o ${tag}_$pm -- python bench.py --steps 2 --warmup 1 --no-graphs $BARGS > $out/pmc_$pm.log 2>&1
done
python - $out $tag <<'PY'
import csv, sys, json, collections, re
out, tag = sys.argv[1], sys.argv[2]
res = {}
for pm in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{out}/{tag}_{pm}_counter_collection.csv")):
        if r["Counter_Name"] != pm: continue
        n = r["Kernel_Name"]
        m = re.search(r"gemm_conv_kernel<([^>]*)>", n)
        key = ("conv" if m.group(1).split(",")[5].strip() == "1" else "gemm") if m else "gemm" if ("gemm_qattn_kernel" in n or "gemm_w22_kernel" in n) else "conv" if "conv_halo_kernel" in n else "attn" if ("attn_fwd" in n or "attn_small" in n) else "norm" if "gn_" in n else None
        if key:
            agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
    res[pm] = {k: {"launches": v[0], "sum_kb": v[1], "avg_kb_per_launch": v[1] / max(1, v[0])} for k, v in agg.items()}
# guide (MI355X_MICROARCH.md, HBM): FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced reads -> x2; KB units
summary = {}
for k in res["FETCH_SIZE"]:
    f = res["FETCH_SIZE"][k]["avg_kb_per_launch"]; w = res["WRITE_SIZE"].get(k, {"avg_kb_per_launch": 0})["avg_kb_per_launch"]
    summary[k] = {"fetch_kb_raw": f, "write_kb_raw": w, "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024,
                  "launches_sampled": res["FETCH_SIZE"][k]["launches"]}
json.dump(summary, open(f"{out}/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(summary))
PY
timeout 240 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $out -o ${tag}_MFMA -- python bench.py --steps 2 --warmup 1 --no-graphs $BARGS > $out/pmc_MFMA.log 2>&1
python - $out $tag <<'PY'
import csv, sys, json, collections, re
out, tag = sys.argv[1], sys.argv[2]
dur = {}
try:
    for r in csv.DictReader(open(f"{out}/{tag}_MFMA_kernel_trace.csv")):
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
except Exception as e:
    print("no kernel trace next to the counters:", e)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f"{out}/{tag}_MFMA_counter_collection.csv")):
    if r["Counter_Name"] != "MfmaUtil": continue
    n = r["Kernel_Name"]
    m = re.search(r"gemm_conv_kernel<([^>]*)>", n)
    key = ("gemm<" if m and m.group(1).split(",")[5].strip() == "0" else "conv<") + m.group(1).replace(" ", "") + ">" if m else "gemm<qattn:64x320,to_q+cross-attention>" if "gemm_qattn_kernel" in n else "gemm<w22:128x160,2x2>" if "gemm_w22_kernel" in n else "conv<halo:4x32px,160>" if "conv_halo_kernel" in n else "attn_fwd" if "attn_fwd" in n else "attn_small" if "attn_small" in n else None
    if key: agg[key].append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 1.0)))
cls = collections.defaultdict(list)
for k, v in agg.items(): cls[k.split("<")[0]] += v
wavg = lambda v: sum(u * d for u, d in v) / sum(d for _u, d in v)
res = {"per_kernel_class_time_weighted_percent": {k: wavg(v) for k, v in cls.items()},
       "per_instance": {k: {"launches": len(v), "time_weighted_percent": wavg(v), "max_percent": max(u for u, _d in v),
                            "total_ms": sum(d for _u, d in v) / 1e6} for k, v in sorted(agg.items())},
       "note": "rocprofv3 --pmc MfmaUtil (SQ_VALU_MFMA_BUSY_CYCLES summed / (GRBM_GUI_ACTIVE x SIMDs)); eager pass of bench.py (--no-graphs, "
               "one launch chain at the full batch B = 4, the launches serialised by the profiler); averages weighted by kernel duration"}
json.dump(res, open(f"{out}/{tag}_mfma_util.json", "w"), indent=1)
print(json.dumps(res["per_kernel_class_time_weighted_percent"]))
PY
rm -f $out/*_kernel_trace.csv $out/*counter_collection.csv $out/*agent_info.csv
# the manifest bench.py reads (copy it with the files into profiles/)
python - $out $tag <<'PY'
import json, os, sys
out, tag = sys.argv[1], sys.argv[2]
man = {"round_tag": tag, "kernel_stats": f"{tag}_kernel_stats.csv", "bench_line": f"{tag}_bench_line.json"}
if os.path.exists(os.path.join(out, f"{tag}_fp8_kernel_stats.csv")): man["fp8_kernel_stats"] = f"{tag}_fp8_kernel_stats.csv"; man["fp8_bench_line"] = f"{tag}_fp8_bench_line.json"
for key, f in (("traffic", f"{tag}_traffic.json"), ("mfma_util", f"{tag}_mfma_util.json")):
    if os.path.exists(os.path.join(out, f)): man[key] = f
json.dump(man, open(os.path.join(out, "MANIFEST.json"), "w"), indent=1)
print(man)
PY
ls $out
