# round 3, call J: what bounds the self-attention tile loop?  ablations: 1 no exp2, 2 no PV MFMAs, 4 no QK^T MFMAs, 8 no LDS-DMA in the loop
for v in base aabl1 aabl2 aabl4 aabl8 aabl6 aabl7 aabl14 aabl15; do
  if [ $v = base ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  echo "== $v: $(python tools/attn_one.py 4 20 1024 1024 2>/dev/null | tail -1) | $(python tools/attn_one.py 4 10 4096 4096 2>/dev/null | tail -1)"
done
