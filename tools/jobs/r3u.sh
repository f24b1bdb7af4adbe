# round 3, call U: ablations of the PIPELINED K loop (1 same tile, 2 no MFMA, 4 no LDS-DMA in the loop, 6 neither)
mkdir -p gpurun_out/r3u; rm -f gpurun_out/r3u/*
L=tools/gemm_lab/lab
for v in new abl1 abl2 abl4 abl6; do
  if [ $v = new ]; then LP=""; else LP=tools/ab/$v; fi
  echo "===== $v" >> gpurun_out/r3u/tl.txt
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br cfgs=12,18,20 reps=20 >> gpurun_out/r3u/tl.txt 2>&1
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,10240,1280,1,g cfgs=14 reps=20 >> gpurun_out/r3u/tl.txt 2>&1
done
python tools/tl_table.py gpurun_out/r3u/tl.txt
