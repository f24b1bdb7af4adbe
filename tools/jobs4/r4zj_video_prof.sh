#!/bin/bash
# round 4: rocprofv3 kernel statistics of the video step alone (complete tile table: no autotune launches in the trace), one chain
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r4zj; mkdir -p $out; rm -rf $out/*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o v1 -- python tools/video_bench.py --steps 10 --streams 1 > $out/line.json 2> $out/log.txt
f=$(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1); echo $f
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per 13 steps-ish:", tot / 1e6)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms {float(r["Percentage"]):5.1f}% n={r["Calls"]:>6s} avg={float(r["AverageNs"]) / 1e3:8.1f}us  {r["Name"][:110]}')
PY
rm -f $out/*kernel_trace.csv $out/*/*kernel_trace.csv
