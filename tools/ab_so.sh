#!/bin/bash
# run a script once per candidate build of libtmix_hip.so: tools/ab_so.sh "<cmd>" a.so b.so ...
cmd=$1; shift
cp tweediemix_amd/lib/libtmix_hip.so /tmp/orig.so
for so in "$@"; do cp $so tweediemix_amd/lib/libtmix_hip.so; echo "== $so: $($cmd 2>&1 )"; done
cp /tmp/orig.so tweediemix_amd/lib/libtmix_hip.so
