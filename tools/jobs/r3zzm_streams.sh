# round 3, call ZZM: one launch chain at B = 4 against two chains of B = 2 on two streams, with this round's (shorter) launches
for s in 1 2 1 2; do
  timeout 600 python bench.py --kind lora --streams $s --no-cpu-baseline --no-trajectory --no-video --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done
