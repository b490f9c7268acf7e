#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_run4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > $O/bench.json; cat $O/bench.json
