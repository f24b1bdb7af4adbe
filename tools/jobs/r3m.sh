for v in base aabl1 aabl2 aabl4 aabl8 aabl16 aabl6 aabl14 aabl30 aabl31; do
  if [ $v = base ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  echo "== $v: $(python tools/attn_one.py 4 20 1024 1024 2>/dev/null | tail -1) | $(python tools/attn_one.py 4 10 4096 4096 2>/dev/null | tail -1)"
done
