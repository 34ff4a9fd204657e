cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 1500 python -m pytest tests -m gpu -q 2>&1 > gpurun_out/r3b/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3b/smoke.log 2>&1
tail -5 gpurun_out/r3b/smoke.log
tail -30 gpurun_out/r3b/pytest.log
