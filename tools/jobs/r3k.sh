for v in pipe0 new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  for i in 1 2; do echo "== $v: $(python tools/attn_one.py 4 20 1024 1024 2>/dev/null | tail -1) | $(python tools/attn_one.py 4 10 4096 4096 2>/dev/null | tail -1)"; done
done
unset TMIX_LIB
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_attention_golden_gpu.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -2
