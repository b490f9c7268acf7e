#!/bin/bash
# Where p2_gen8_kernel's time goes: ablation builds (results wrong by construction, only kernel times are read).
#   variant 1 = no m-contraction, 2 = no n-contraction, 3 = neither, 7 = neither and no Psi1 product (k-loop only)
#   tools/r03_gen8_ablate.sh build ; gpurun -- 'bash tools/r03_gen8_ablate.sh run'
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" == "build" ]; then
  cp "$ROOT/gparml_amd/libgparml_hip.so" "$ROOT/gparml_amd/lib_base.so.bin"
  for v in 1 2 3 7; do
    GPARML_OBJ_TAG=_g8abl$v GPARML_EXTRA_FLAGS="-DGPARML_GEN8_ABLATE=$v" GPARML_LIB_OUT="$ROOT/gparml_amd/lib_g8abl$v.so.bin" "$ROOT/tools/build_lib.sh" | tail -1
  done
else
  O=$ROOT/gpurun_out/r03_gen8_ablate; mkdir -p $O
  cd /tmp && export TMPDIR=/tmp
  for v in base g8abl1 g8abl2 g8abl3 g8abl7; do
    export GPARML_LIB=$ROOT/gparml_amd/lib_$v.so.bin
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/$v -o $v --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --N 100000 --D 100 --M 512 --Q 10 --regime B > $O/$v.log 2>&1
    python3 - <<PY
import csv, glob
for f in glob.glob('$O/$v/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'gen8' in r['Name'] or 'point_kernel' in r['Name']: print('%-8s %-48s avg_us=%9.1f' % ('$v', r['Name'][:48], float(r['AverageNs'])/1e3))
PY
    rm -rf $O/$v/*kernel_trace.csv $O/$v/*agent_info.csv
  done
fi
