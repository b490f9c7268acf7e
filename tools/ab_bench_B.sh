#!/bin/bash
# same-box A/B for the free-embedding (regime B) workload at config-3 size: see tools/ab_bench.sh
set -u
cd "$(dirname "$0")/.."
for r in 1 2; do
  for v in "$@"; do
    GPARML_LIB=$PWD/gparml_amd/lib_$v.so.bin python3 bench.py --steps 3 --warmup 1 --regime B --no-cpu-baseline | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'], 2), 'F', d['config']['F'])"
  done
done
