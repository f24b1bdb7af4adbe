#!/bin/bash
# round 5: the whole -m gpu suite + smoke() on the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/r5s/tests.log 2>&1; tail -4 gpurun_out/r5s/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5s/smoke.log 2>&1; tail -1 gpurun_out/r5s/smoke.log
