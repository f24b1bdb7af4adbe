# round 3, call O: register-staged tiling 21 (check + timelines), the three tile tables on one box, the LoRA small-delta test
mkdir -p gpurun_out/r3o; rm -f gpurun_out/r3o/*
L=tools/gemm_lab/lab
timeout 300 $L check nocold 512,512,256,1,b 1024,1280,1280,1,br 300,264,128,1,b 2048,2560,1280,1,brs 4096,1280,320,1,br 520,640,64,1,br 520,640,128,1,br 520,640,192,1,br cfgs=21 reps=3 2>&1 | grep -E "check|rc" 
echo "===== new" > gpurun_out/r3o/tl.txt
timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br 2048,1280,1280,1,br 16384,640,2560,1,br cfgs=12,20,21 reps=20 >> gpurun_out/r3o/tl.txt 2>&1
python tools/tl_table.py gpurun_out/r3o/tl.txt
timeout 300 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "small_lora" 2>&1 | grep -E "delta contribution|passed|failed"
for i in 1 2; do
for tb in shipped gpurun_out/r3g/tuned_quick.json gpurun_out/r3n/tuned_r3.json; do
  if [ $tb = shipped ]; then unset TMIX_TUNE_FILE; else export TMIX_TUNE_FILE=$tb; fi
  timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done
