# round 3, call ZZZ: loader waves at issue priority 3 against the default 0 (they share a SIMD with a math wave), same box
for r in 1 2 3; do
for v in base lprio3; do
  if [ $v = base ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done
