#!/bin/bash
# dev: build a variant of libtmix_hip.so into tools/ab/<name>/ (git-ignored; travels to the GPU box with gpurun):
#   tools/build_variant.sh abl1 -DTMIX_ABL=1
# use it with  LD_LIBRARY_PATH=tools/ab/<name> tools/gemm_lab/lab ...   or  TMIX_LIB=tools/ab/<name>/libtmix_hip.so python ...
name=$1; shift
cd "$(dirname "$0")/../tweediemix_amd/csrc" && make -j8 EXTRA="$*" OBJDIR=../../build/obj_$name OUT=../../tools/ab/$name/libtmix_hip.so 2>&1 | grep -E "error|Error" ; ls -la ../../tools/ab/$name/libtmix_hip.so
