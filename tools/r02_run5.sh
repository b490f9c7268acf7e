#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_run5
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -5 $O/pytest1.log
timeout 300 python bench.py --steps 3 --warmup 1 --N 30000 --D 12 --M 96 --Q 5 --no-cpu-baseline > $O/small.log 2>&1; echo "small rc=$?"; tail -3 $O/small.log
