cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v Warn | tail -6
python tools/tolerance_audit.py 2>&1 | grep "looser\|bounds," | tail -30
for st in stage2 stage3; do python tools/train_bench.py $st 10 2>&1 | tail -1; python tools/train_bench.py $st 10 graph 2>&1 | tail -1; done
