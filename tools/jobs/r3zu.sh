# round 3, call ZU: LoRA in its low-rank form: kernel test, projection-level delta test, tiny-UNet parity vs oracle and vs the merged plan
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "lora_down" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "low_rank" 2>&1 | tail -12
