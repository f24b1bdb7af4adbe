#!/bin/bash
# round 6: (a) convolution K order tap-major (tools/ab/head = the last commit) vs channel-chunk major (working tree), (b) the halo-patch kernel on the 64^2 / 32^2 convolutions;
# same box, the captured step, then FETCH_SIZE per convolution launch under rocprofv3 for the four variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6e; mkdir -p $out
python tools/table_with_halo.py tweediemix_amd/tuned_gfx950.json $out/halo_small.json --small-tiles
run() {   # name lib table
  export TMIX_LIB=$2 TMIX_TUNE_FILE=$3
  python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline > $out/bench_$1.json 2> $out/bench_$1.err
  python - $out/bench_$1.json $1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["roofline"]["classes"]
print("%-22s ms_per_step %.3f | conv %.3f ms (%.0f TF) gemm %.3f attn %.3f norm %.3f | boundaries %.3f | sclk %s W %s" % (sys.argv[2], d["ms_per_step"], c["conv"]["sum_launch_ms"], c["conv"]["tflops"],
      c["gemm"]["sum_launch_ms"], c["attn"]["sum_launch_ms"], c["norm"]["sum_launch_ms"], d["roofline"]["kernel_boundaries_ms"], d["chip_state_under_load"].get("sclk_mhz"), d["chip_state_under_load"].get("power_raw")), flush=True)
PY
}
for rep in 1 2; do
  run tapmajor_shipped_$rep  tools/ab/head/libtmix_hip.so tweediemix_amd/tuned_gfx950.json
  run tapmajor_halo_$rep     tools/ab/head/libtmix_hip.so $out/halo_small.json
  run chunkmajor_shipped_$rep tweediemix_amd/lib/libtmix_hip.so tweediemix_amd/tuned_gfx950.json
  run chunkmajor_halo_$rep    tweediemix_amd/lib/libtmix_hip.so $out/halo_small.json
done
for v in tapmajor_shipped tapmajor_halo chunkmajor_shipped chunkmajor_halo; do
  case $v in tapmajor*) export TMIX_LIB=tools/ab/head/libtmix_hip.so;; *) export TMIX_LIB=tweediemix_amd/lib/libtmix_hip.so;; esac
  case $v in *halo) export TMIX_TUNE_FILE=$out/halo_small.json;; *) export TMIX_TUNE_FILE=tweediemix_amd/tuned_gfx950.json;; esac
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc -o $v -- python bench.py --steps 2 --warmup 1 --no-graphs --kind lora --no-trajectory --no-cpu-baseline --no-video > $out/pmc_$v.log 2>&1
  python - $out/pmc/${v}_counter_collection.csv $v <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    n = r["Kernel_Name"]
    m = re.search(r"gemm_conv_kernel<([^>]*)>", n)
    if m and m.group(1).split(",")[5].strip() == "1": key = "conv<" + ",".join(x.strip() for x in m.group(1).split(",")[:5]) + ",sc=" + m.group(1).split(",")[-1].strip() + ">"
    elif "conv_halo_kernel" in n: key = "conv<halo 4x32>"
    else: continue
    agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
tot_n = sum(v[0] for v in agg.values()); tot = sum(v[1] for v in agg.values())
print(f"{sys.argv[2]}: FETCH_SIZE per conv launch (x2 = bytes, gfx950 correction): {2 * tot / tot_n / 1024:.1f} MB over {tot_n} launches; " +
      " ".join(f"{k}: {2 * v[1] / v[0] / 1024:.0f} MB x{v[0]}" for k, v in sorted(agg.items())), flush=True)
PY
done
rm -rf $out/pmc
