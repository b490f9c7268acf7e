#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_run7
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hp_truth.py tests/test_gpu_fullsize.py tests/test_gpu_partial_terms.py tests/test_gpu_global_step.py tests/test_gpu_pipeline.py tests/test_gpu_dropout.py -m gpu -x -q -k "not config4" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2', d['ms_per_step'], d['config']['device_ms'], d['config']['F'])"
python3 bench.py --steps 50 --warmup 5 --N 100000 --D 10 --M 128 --Q 10 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config1', d['ms_per_step'], d['config']['device_ms'])"
