# round 3, call L: pipelined self-attention kernel vs the round-2 one (TMIX_ATTN_OLD=1), correctness tests first
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_attention_golden_gpu.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -2
for i in 1 2; do
echo "== old: $(TMIX_ATTN_OLD=1 python tools/attn_one.py 4 20 1024 1024 2>/dev/null | tail -1) | $(TMIX_ATTN_OLD=1 python tools/attn_one.py 4 10 4096 4096 2>/dev/null | tail -1) | $(TMIX_ATTN_OLD=1 python tools/attn_one.py 2 20 1024 1024 2>/dev/null | tail -1)"
echo "== new: $(python tools/attn_one.py 4 20 1024 1024 2>/dev/null | tail -1) | $(python tools/attn_one.py 4 10 4096 4096 2>/dev/null | tail -1) | $(python tools/attn_one.py 2 20 1024 1024 2>/dev/null | tail -1)"
done
