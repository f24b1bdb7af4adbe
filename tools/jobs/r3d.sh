timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "wide_epilogue" 2>&1 | grep -E "assert|Error|FAILED|passed|failed|Mismatch|mismatch|name" | head -30
