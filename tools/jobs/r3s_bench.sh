# round 3, call S: the default bench line (what the driver runs) with the shipped library and table; the LoRA merge test
mkdir -p gpurun_out/r3s
timeout 300 python -m pytest tests/test_unet_gpu.py -m gpu -q -s -k "lora_delta" 2>&1 | grep -E "delta contribution|passed|failed"
timeout 1500 python bench.py > gpurun_out/r3s/bench.json 2> gpurun_out/r3s/bench.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r3s/bench.json)"
