# round 3, call ZI: one-transcendental GELU + GEGLU bias from LDS, host-computed prefetch split, loaders hand tile 0 over after two stages: lab check / timeline, A/B vs the previous commit (tools/ab/head), kernel tests
mkdir -p gpurun_out/r3zi; rm -f gpurun_out/r3zi/*
L=tools/gemm_lab/lab
timeout 600 $L check nocold 1024,2560,640,1,g 2048,10240,1280,1,g 512,512,256,1,g 1024,1280,1280,1,br 300,264,128,1,b 2048,2560,1280,1,brs 520,648,64,1,brs 520,640,64,1,br 520,640,128,1,br 520,640,192,1,br cfgs=1,2,3,4,5,7,12,13,14,15,16,17,18,19,20,21 reps=3 > gpurun_out/r3zi/check.txt 2>&1
grep -c " ok" gpurun_out/r3zi/check.txt; grep "WRONG\|rc " gpurun_out/r3zi/check.txt | head
for v in head new; do
  if [ $v = new ]; then LP=""; else LP=tools/ab/$v; fi
  echo "===== $v" >> gpurun_out/r3zi/tl.txt
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,10240,1280,1,g 16384,5120,640,1,g cfgs=14,4 reps=20 >> gpurun_out/r3zi/tl.txt 2>&1
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br cfgs=20,21 reps=20 >> gpurun_out/r3zi/tl.txt 2>&1
done
python tools/tl_table.py gpurun_out/r3zi/tl.txt
for i in 1 2; do
for v in head new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
unset TMIX_LIB
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_text_gpu.py -m gpu -q -x 2>&1 | tail -3
