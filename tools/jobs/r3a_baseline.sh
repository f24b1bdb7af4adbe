# round 3, call A: where the shipped (round-2) library stands -- per-workgroup timelines of the step's main GEMM shapes,
# and the same shapes through torch/hipBLASLt (hot loop and cold-weights), written as JSON for profiles/r3_vs_blas.json
mkdir -p gpurun_out/r3a
L=tools/gemm_lab/lab
timeout 900 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br 4096,3840,1280,1,b 4096,10240,1280,1,g 16384,640,640,1,br 16384,5120,640,1,g 2048,1280,1280,1,br cfgs=4,7,12,14,16,17,18 reps=20 > gpurun_out/r3a/tl.txt 2>&1
timeout 600 python tools/vs_blas.py gpurun_out/r3a/vs_blas.json > gpurun_out/r3a/vs_blas.txt 2>&1
timeout 600 python tools/cold_vs_blas.py gpurun_out/r3a/cold_vs_blas.json > gpurun_out/r3a/cold_vs_blas.txt 2>&1
tail -5 gpurun_out/r3a/vs_blas.txt gpurun_out/r3a/cold_vs_blas.txt
