#!/bin/bash
# Same-box A/B of library builds or run-time switches: the pool's boxes differ by +-2 %, so a 1 % kernel change is only visible when the variants run
# alternately on ONE box.  Replaces ab_bench.sh / ab_bench_B.sh / ab_bench_c4.sh and the r0x ablation one-offs (git history).
#   build a variant:   GPARML_EXTRA_FLAGS=-DGPARML_X GPARML_LIB_OUT=$PWD/gparml_amd/lib_NAME.so.bin tools/build_lib.sh   (git-ignored, travels with gpurun)
#   run:               gpurun -- 'tools/ab.sh [-r ROUNDS] [-a "bench.py arguments"] NAME[:ENV=VALUE[,ENV=VALUE]] ...'
#                      NAME = "intree" (gparml_amd/libgparml_hip.so) or a lib_NAME.so.bin; the optional ENV list sets run-time switches (GPARML_DD_KIPSI2=0 ...)
# prints ms per evaluation and the stage / kernel times of every variant, ROUNDS rounds interleaved.
set -u
cd "$(dirname "$0")/.."
ROUNDS=3; ARGS="--steps 20 --warmup 3"
while getopts "r:a:" o; do case $o in r) ROUNDS=$OPTARG;; a) ARGS=$OPTARG;; esac; done
shift $((OPTIND - 1))
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""
    [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
    lib=$PWD/gparml_amd/lib_$v.so.bin; [ "$v" == "intree" ] && lib=$PWD/gparml_amd/libgparml_hip.so
    env $envs GPARML_LIB=$lib python3 bench.py $ARGS --no-cpu-baseline --no-extra | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['config'].get('device_ms') or d['config'].get('kernel_ms')
print('$spec', round(d['ms_per_step'], 3), 'p2', k['p2_kernel_ms'], 'p1', k['p1_kernel_ms'], 'psi1', k['psi1_ms'], 'global', k['global_ms'], 'total', k['total_ms'])"
  done
done
