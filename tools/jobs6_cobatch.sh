#!/bin/bash
# round 6, VERDICT r5 item 4: kernel evidence for the co-batched plans behind images/s (8 seeds per launch: fusion / start B = 32, plain B = 16)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6g; mkdir -p $out
for k in fusion plain start; do
  python tools/step_shapes.py $k --seeds-per-gpu 8 --kind lora > $out/cobatch8_${k}_shapes.txt 2>&1
  tail -1 $out/cobatch8_${k}_shapes.txt
done
python tools/vs_blas.py --cobatch $out/cobatch_vs_blas_hot.json > $out/cobatch_vs_blas_hot.txt 2>&1; cat $out/cobatch_vs_blas_hot.txt
timeout 600 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $out/pmc -o mfma -- python bench.py --seeds-per-gpu 8 --steps 2 --warmup 1 --no-graphs --kind lora --no-trajectory --no-cpu-baseline --no-video > $out/pmc_mfma.log 2>&1
python - $out <<'PY'
import csv, sys, json, collections, re
out = sys.argv[1]
dur = {}
for r in csv.DictReader(open(f"{out}/pmc/mfma_kernel_trace.csv")):
    dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f"{out}/pmc/mfma_counter_collection.csv")):
    if r["Counter_Name"] != "MfmaUtil": continue
    n = r["Kernel_Name"]
    m = re.search(r"gemm_conv_kernel<([^>]*)>", n)
    key = ("gemm<" if m and m.group(1).split(",")[5].strip() == "0" else "conv<") + m.group(1).replace(" ", "") + ">" if m else "gemm<qattn>" if "gemm_qattn_kernel" in n else "conv<halo>" if "conv_halo_kernel" in n else "attn_fwd" if "attn_fwd" in n else "attn_small" if "attn_small" in n else None
    if key: agg[key].append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 1.0)))
cls = collections.defaultdict(list)
for k, v in agg.items(): cls[k.split("<")[0]] += v
wavg = lambda v: sum(u * d for u, d in v) / sum(d for _u, d in v)
res = {"workload": "bench.py --seeds-per-gpu 8 (fusion plan, B = 32), eager pass under rocprofv3 --pmc MfmaUtil",
       "per_kernel_class_time_weighted_percent": {k: wavg(v) for k, v in cls.items()},
       "per_instance": {k: {"launches": len(v), "time_weighted_percent": wavg(v), "total_ms": sum(d for _u, d in v) / 1e6} for k, v in sorted(agg.items())}}
json.dump(res, open(f"{out}/cobatch8_mfma_util.json", "w"), indent=1)
print(json.dumps(res["per_kernel_class_time_weighted_percent"]))
PY
rm -rf $out/pmc
