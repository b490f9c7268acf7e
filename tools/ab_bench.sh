#!/bin/bash
# Same-box A/B of library builds: the pool's boxes differ by +-2 %, so a 1 % kernel change is only visible when the variants run
# alternately on ONE box.  Build each variant, copy it to gparml_amd/lib_<name>.so.bin (git-ignored, travels with gpurun; selected through GPARML_LIB, the in-tree
# library is not touched), then
#   gpurun -- 'bash tools/ab_bench.sh base variant1 variant2'
# prints ms per evaluation and the phase-2 / phase-1 kernel times of every variant, three rounds interleaved.
set -u
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for v in "$@"; do
    GPARML_LIB=$PWD/gparml_amd/lib_$v.so.bin python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['device_ms']
print('$v', round(d['ms_per_step'], 3), 'p2', k['p2_kernel_ms'], 'p1', k['p1_kernel_ms'], 'psi1', k['psi1_ms'], 'global', k['global_ms'])"
  done
done
