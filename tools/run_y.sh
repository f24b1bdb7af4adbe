timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "groupnorm" 2>&1 | tail -2
echo "== prev"; (cd tools/ab/prev && timeout 100 python tools/gn_time.py 2>&1 | grep groupnorm | head -9)
echo "== new"; timeout 100 python tools/gn_time.py 2>&1 | grep groupnorm | head -9
