# round 3, call ZZX: does the timed region's length change ms_per_step?  (default 20 steps / 3 warm-up against longer runs, same box)
for a in "--steps 20 --warmup 3" "--steps 40 --warmup 3" "--steps 50 --warmup 10" "--steps 100 --warmup 10" "--steps 20 --warmup 3" "--steps 50 --warmup 10"; do
  timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', round(d['value'],2), round(d['ms_per_step'],3))"
done
