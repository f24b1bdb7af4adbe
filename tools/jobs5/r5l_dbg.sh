#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
export TMIX_LIB=$PWD/tools/ab/xcdx/libtmix_hip.so
timeout 600 python -m pytest tests/test_sampler_gpu.py -x -q -k "two_seeds" > gpurun_out/r5l/a.log 2>&1; tail -3 gpurun_out/r5l/a.log
cat > /tmp/no23.py <<'PY'
import tweediemix_amd.lib as L
L.TILE_CANDIDATES = tuple(c for c in L.TILE_CANDIDATES if c != 23)
import pytest, sys
sys.exit(pytest.main(["tests/test_sampler_gpu.py", "-x", "-q", "-k", "two_seeds"]))
PY
timeout 600 python /tmp/no23.py > gpurun_out/r5l/b.log 2>&1; tail -3 gpurun_out/r5l/b.log
unset TMIX_LIB
timeout 600 python -m pytest tests/test_sampler_gpu.py -x -q -k "two_seeds" > gpurun_out/r5l/c.log 2>&1; tail -3 gpurun_out/r5l/c.log
