mkdir -p gpurun_out
timeout 900 python tools/plain_time.py 2>&1 | grep -v amdgpu | tail -8
timeout 900 python tools/video_bench.py --streams 2 2>&1 | tail -3
timeout 900 python tools/video_bench.py --streams 1 2>&1 | tail -3
