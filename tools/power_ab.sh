#!/bin/bash
# Where does the power go?    Socket power, shader clock and joules per evaluation of the headline loop for library variants on ONE
# box: the product build against builds whose phase-2 kernel stages nothing in its k-loop (no HBM / L2 -> LDS traffic there; results wrong by
# construction).  usage: gpurun -- 'tools/power_ab.sh base abl2'   (variants = gparml_amd/lib_NAME.so.bin)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  GPARML_LIB=$PWD/gparml_amd/lib_$v.so.bin tools/clock_probe.sh gpurun_out/power_$v.txt python3 bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-extra > /dev/null 2>&1
  echo "== $v"; grep "loaded samples" gpurun_out/power_$v.txt
done
