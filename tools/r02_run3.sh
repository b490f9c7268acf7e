#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/r02_run3
mkdir -p $O
cd $R
for v in "GP_P2_STAGGER=0" "GP_P2_STAGGER=1" "GP_P2_STAGGER=2" "GP_P2_STAGGER=0" "GP_P2_STAGGER=1"; do
  echo "== $v" >> $O/ablate.log
  env $v python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['device_ms'])" >> $O/ablate.log
done
cat $O/ablate.log
