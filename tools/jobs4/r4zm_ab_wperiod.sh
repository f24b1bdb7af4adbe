#!/bin/bash
# round 4: same-box A/B of the headline step: the library before tmix_gemm_desc.w_period (0506c0c, tools/ab/head) against the current one
mkdir -p gpurun_out/r4zm; rm -f gpurun_out/r4zm/*
for r in 1 2 3; do for v in head cur; do
  if [ $v = head ]; then L=tools/ab/head/libtmix_hip.so; else L=tweediemix_amd/lib/libtmix_hip.so; fi
  ms=$(TMIX_LIB=$L timeout 600 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["classes"]["gemm"]["sum_launch_ms"])')
  echo "$v $ms"; done; done | tee gpurun_out/r4zm/ab.txt
