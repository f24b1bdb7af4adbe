# round 3, call ZZH: do the launches wait for instruction fetch?  I-cache requests / misses and the mean instruction-fetch latency per kernel (eager pass of the bench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3zzh; rm -rf $out; mkdir -p $out
BARGS="--kind lora --no-trajectory --no-cpu-baseline --no-video --steps 2 --warmup 1 --no-graphs"
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $out -o ic -- python bench.py $BARGS > $out/ic.log 2>&1
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out -o ifl -- python bench.py $BARGS > $out/ifl.log 2>&1
python - $out <<'PY'
import csv, sys, collections, re, json
out = sys.argv[1]
def load(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f"{out}/{tag}_counter_collection.csv")):
        k = r["Kernel_Name"]
        m = re.search(r"gemm_conv_kernel<([^>]*)>", k)
        key = "gemm<" + m.group(1).replace(" ", "") + ">" if m else k.split("(")[0][-40:]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[key] += 1
    return agg, n
try:
    a, n = load("ic")
    print("kernel, launches, icache req / launch, misses / launch, dup misses / launch, miss rate")
    for k in sorted(a, key=lambda k: -a[k]["SQC_ICACHE_REQ"])[:14]:
        v = a[k]; print(k, n[k], round(v["SQC_ICACHE_REQ"] / n[k]), round(v["SQC_ICACHE_MISSES"] / n[k]), round(v["SQC_ICACHE_MISSES_DUPLICATE"] / n[k]), round(v["SQC_ICACHE_MISSES"] / max(1, v["SQC_ICACHE_REQ"]), 3))
except Exception as e: print("ic pass:", e)
try:
    a, n = load("ifl")
    print("kernel, launches, ifetch / launch, mean fetch latency (cycles), wave cycles / launch, wait-inst-any / wave cycles")
    for k in sorted(a, key=lambda k: -a[k]["SQ_WAVE_CYCLES"])[:14]:
        v = a[k]; print(k, n[k], round(v["SQ_IFETCH"] / n[k]), round(v["SQ_IFETCH_LEVEL"] / max(1, v["SQ_IFETCH"]), 1), round(v["SQ_WAVE_CYCLES"] / n[k]), round(v["SQ_WAIT_INST_ANY"] / max(1, v["SQ_WAVE_CYCLES"]), 3))
except Exception as e: print("ifl pass:", e)
PY
rm -f $out/*_kernel_trace.csv $out/*agent_info.csv
