#!/bin/bash
# round 4: same-box A/B of the headline step: before w_period (tools/ab/head), w_period fields at the end of Params (tools/ab/cur_libtmix_hip.so), beside M/N/K (working tree)
mkdir -p gpurun_out/r4zn; rm -f gpurun_out/r4zn/*
for r in 1 2 3; do for v in head cur new; do
  case $v in head) L=tools/ab/head/libtmix_hip.so;; cur) L=tools/ab/cur_libtmix_hip.so;; new) L=tweediemix_amd/lib/libtmix_hip.so;; esac
  ms=$(TMIX_LIB=$L timeout 600 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["classes"]["gemm"]["sum_launch_ms"])')
  echo "$v $ms"; done; done | tee gpurun_out/r4zn/ab.txt
