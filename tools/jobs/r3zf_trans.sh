# round 3, call ZF: what the transposed V region costs the QKV launch (flag t: last third of N stored transposed), per tiling
mkdir -p gpurun_out/r3zf; rm -f gpurun_out/r3zf/*
L=tools/gemm_lab/lab
echo "===== new" >> gpurun_out/r3zf/tl.txt
timeout 300 $L tl 1024,3840,1280,4,b 1024,3840,1280,4,bt cfgs=16,2,12,21,17 reps=20 >> gpurun_out/r3zf/tl.txt 2>&1
cat gpurun_out/r3zf/tl.txt | cut -c1-200
