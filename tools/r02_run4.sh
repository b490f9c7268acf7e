#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_run4
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hp_truth.py tests/test_gpu_fullsize.py tests/test_gpu_partial_terms.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > $O/bench.json; cat $O/bench.json
