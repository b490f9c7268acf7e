#!/bin/bash
# full GPU suite + default bench + configs[1] bench
set -u
R=${GRAFT_REPO_ROOT:?}
cd $R
mkdir -p gpurun_out/r02_run11
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_run11/pytest.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r02_run11/pytest.log
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_run11/bench.json; python3 -c "
import json; d=json.load(open('gpurun_out/r02_run11/bench.json')); print(d['value'], d['ms_per_step'], d['config']['device_ms'], d['roofline'])"
python3 bench.py --steps 50 --warmup 5 --N 100000 --D 10 --M 128 --Q 10 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_run11/config1_A.json; python3 -c "
import json; d=json.load(open('gpurun_out/r02_run11/config1_A.json')); print('config1 A', d['ms_per_step'], d['config']['device_ms'])"
python3 bench.py --steps 20 --warmup 3 --N 100000 --D 10 --M 128 --Q 10 --regime B --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_run11/config1_B.json; python3 -c "
import json; d=json.load(open('gpurun_out/r02_run11/config1_B.json')); print('config1 B', d['ms_per_step'], d['config']['device_ms'])"
