timeout 300 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "small_lora" 2>&1 | tail -15
for i in 1 2; do
for tb in shipped tools/ab/tables/quick.json tools/ab/tables/r3.json; do
  if [ $tb = shipped ]; then unset TMIX_TUNE_FILE; else export TMIX_TUNE_FILE=$tb; fi
  timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done
