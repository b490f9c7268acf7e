#!/bin/bash
# full GPU suite, default bench, round-2 profile of the current kernels
set -u
R=${GRAFT_REPO_ROOT:?}
cd $R
mkdir -p gpurun_out/r02_run9
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_run9/pytest.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r02_run9/pytest.log
python3 bench.py --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/r02_run9/bench.json; cat gpurun_out/r02_run9/bench.json
bash tools/r02_prof.sh p1v2 2>&1 | tail -40
