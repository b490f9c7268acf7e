#!/bin/bash
# numbers quoted in DESIGN.md: device errors against the extended-precision truth, configs[1] timings
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_numbers
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_hp_truth.py -m gpu -q -s 2>&1 | grep "error vs truth" > $O/hp_truth.txt; cat $O/hp_truth.txt
python3 bench.py --steps 50 --warmup 5 --N 100000 --D 10 --M 128 --Q 10 --no-cpu-baseline 2>&1 | tail -1 > $O/config1_A.json
python3 bench.py --steps 20 --warmup 3 --N 100000 --D 10 --M 128 --Q 10 --regime B --no-cpu-baseline 2>&1 | tail -1 > $O/config1_B.json
python3 bench.py --steps 5 --warmup 1 --regime B --no-cpu-baseline 2>&1 | tail -1 > $O/config2_B.json
python3 bench.py --steps 20 --warmup 3 2>&1 | tail -1 > $O/bench_default.json
for f in config1_A config1_B config2_B bench_default; do python3 -c "
import json,sys; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],4), d['config']['device_ms'], d.get('cpu_baseline'))"; done
