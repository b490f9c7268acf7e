#!/bin/bash
# r04: per-seed error floor of grad_Z (device statistics / exact statistics through global steps of increasing precision), N = 1e5
mkdir -p gpurun_out
for sz in "100" "101 11" "102 12" "103 13" "104 14"; do
  python tests/devtools/dev_refine_with_gpu_stats.py ${1:-100000} $sz 2>&1 | grep -v Warning
done > gpurun_out/r04_seed_floor_${1:-100000}.txt 2>&1
tail -70 gpurun_out/r04_seed_floor_${1:-100000}.txt
