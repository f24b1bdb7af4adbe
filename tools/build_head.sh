#!/bin/bash
# dev: build libtmix_hip.so of the last COMMIT into tools/ab/head/ (for same-box A/B of the working tree against it):  tools/build_head.sh [rev]
rev=${1:-HEAD}
root="$(cd "$(dirname "$0")/.." && pwd)"
t=/tmp/tmix_head_src; rm -rf $t; mkdir -p $t/csrc $t/include
for f in $(git -C $root ls-tree --name-only $rev tweediemix_amd/csrc/); do git -C $root show $rev:$f > $t/csrc/$(basename $f); done
git -C $root show $rev:include/tmix.h > $t/include/tmix.h
sed -i "s|../../include/tmix.h|$t/include/tmix.h|" $t/csrc/common.h $t/csrc/Makefile
mkdir -p $root/tools/ab/head
(cd $t/csrc && make -j8 OBJDIR=$t/obj OUT=$root/tools/ab/head/libtmix_hip.so 2>&1 | grep -E "error|Error"); ls -la $root/tools/ab/head/libtmix_hip.so
