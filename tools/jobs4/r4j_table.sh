#!/bin/bash
# round 4: conv launches with shortcut taps get their own tile-table entries (the key now carries the shortcut channels)
mkdir -p gpurun_out/r4j
timeout 1500 python tools/extend_table_missing.py gpurun_out/r4j/tuned_gfx950.json > gpurun_out/r4j/extend.log 2>&1; tail -5 gpurun_out/r4j/extend.log
