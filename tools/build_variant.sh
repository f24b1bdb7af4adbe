#!/bin/bash
# dev: build a variant of libtmix_hip.so into tools/ab/<name>/ (git-ignored; travels to the GPU box with gpurun):
#   tools/build_variant.sh abl1 -DTMIX_ABL=1
#   ONLY="attention" tools/build_variant.sh attn_noexp -DTMIX_ATTN_ABL=1     (recompile only that object; the rest is copied from build/obj)
#   EXPERIMENTAL=1 tools/build_variant.sh exp                            (tilings 24 / 25: gemm_ff1p.hip and the L2-prefetcher wave of gemm_w22.hip)
# use it with  LD_LIBRARY_PATH=tools/ab/<name> tools/gemm_lab/lab ...   or  TMIX_LIB=tools/ab/<name>/libtmix_hip.so python ...
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
if [ -n "$ONLY" ]; then
  mkdir -p $root/build/obj_$name $root/tools/ab/$name && cp -p $root/build/obj/*.o $root/build/obj_$name/ && for o in $ONLY; do rm -f $root/build/obj_$name/$o.o; done
fi
cd $root/tweediemix_amd/csrc && make -j8 EXPERIMENTAL=${EXPERIMENTAL:-0} EXTRA="$*" OBJDIR=../../build/obj_$name OUT=../../tools/ab/$name/libtmix_hip.so 2>&1 | grep -E "error|Error" ; ls -la ../../tools/ab/$name/libtmix_hip.so
