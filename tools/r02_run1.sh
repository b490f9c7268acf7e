#!/bin/bash
# round-2 GPU pass 1: full GPU test suite, then the bench with the eight-wave and the four-wave fast phase-2 kernel
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_run1
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_w8.log 2>&1; tail -1 $O/bench_w8.log
GP_P2_VARIANT=4 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_w4.log 2>&1; tail -1 $O/bench_w4.log
