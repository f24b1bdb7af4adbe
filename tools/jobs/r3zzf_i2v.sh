# round 3, call ZZF: the video UNet on the shared ResnetBlock2D emitter (shortcut taps, no concat launches): tests, then the video step old / new
mkdir -p gpurun_out/r3zzf
timeout 1500 python -m pytest tests/test_i2vgen_gpu.py tests/test_video_gpu.py -m gpu -q -x 2>&1 | tail -3
for v in 1 0; do
  if [ $v = 1 ]; then export TMIX_SHORTCUT_GEMM=1; else unset TMIX_SHORTCUT_GEMM; fi
  echo "TMIX_SHORTCUT_GEMM=$v"; timeout 900 python tools/video_bench.py 2>&1 | grep -v amdgpu.ids | tail -2
done
