# round 3, call ZN: the whole -m gpu suite + smoke on the current tree
mkdir -p gpurun_out/r3zn
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r3zn/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3zn/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tail -2
