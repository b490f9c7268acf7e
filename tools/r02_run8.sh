#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
cd $R
for w in 352 352 370; do
GP_P1_WG=$w python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('wG=$w', d['ms_per_step'], d['config']['device_ms']['p1_kernel_ms'])"
done
