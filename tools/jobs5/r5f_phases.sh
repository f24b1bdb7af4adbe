#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
TMIX_TUNE_FILE=$PWD/tweediemix_amd/tuned_gfx950.json timeout 600 python tools/insitu_phases.py lora gemm > gpurun_out/r5f/old.txt 2> gpurun_out/r5f/old.err
TMIX_TUNE_FILE=$PWD/tools/tables/r5d_tuned.json timeout 600 python tools/insitu_phases.py lora gemm > gpurun_out/r5f/new.txt 2> gpurun_out/r5f/new.err
head -14 gpurun_out/r5f/old.txt; head -14 gpurun_out/r5f/new.txt; tail -3 gpurun_out/r5f/new.err
