# round 3, call ZA: ablations of the staged plain epilogue (16 no bias loads, 32 no C stores, 64 no residual, 112 all three)
mkdir -p gpurun_out/r3za; rm -f gpurun_out/r3za/*
L=tools/gemm_lab/lab
for v in new abl16 abl32 abl64 abl112; do
  if [ $v = new ]; then LP=""; else LP=tools/ab/$v; fi
  echo "===== $v" >> gpurun_out/r3za/tl.txt
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br 4096,3840,1280,1,b cfgs=12,21 reps=20 >> gpurun_out/r3za/tl.txt 2>&1
done
python tools/tl_table.py gpurun_out/r3za/tl.txt
