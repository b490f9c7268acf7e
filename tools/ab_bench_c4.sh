#!/bin/bash
# same-box A/B of library builds on the regime-B shapes: configs[4]'s per-GPU shape (2e4-point slice) and configs[2]'s with free embeddings (1e5 points, tile kernel forced)
set -u
cd "$(dirname "$0")/.."
for r in 1 2; do
  for v in "$@"; do
    GPARML_LIB=$PWD/gparml_amd/lib_$v.so.bin python3 bench.py --steps 3 --warmup 1 --N 20000 --D 1000 --M 1024 --Q 50 --regime B --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v c4', round(d['ms_per_step'], 2), 'p2', d['config']['device_ms']['p2_kernel_ms'], 'F', d['config']['F'])"
    GPARML_B_PHASE2=tiles GPARML_LIB=$PWD/gparml_amd/lib_$v.so.bin python3 bench.py --steps 3 --warmup 1 --N 100000 --D 100 --M 512 --Q 10 --regime B --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v c2-tiles', round(d['ms_per_step'], 2), 'p2', d['config']['device_ms']['p2_kernel_ms'], 'F', d['config']['F'])"
  done
done
