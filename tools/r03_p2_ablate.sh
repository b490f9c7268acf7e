#!/bin/bash
# Upper bound on what a 256 x 128 workgroup tile could buy the fast phase-2 kernel, without building it: ablation builds of the library
# that issue 25 % fewer (variant 1) or no (variant 2) LDS-DMA staging instructions in p2_fast8_kernel's k-loop.  Results are wrong by
# construction; only the kernel time is read.  Build here (CPU), run on the GPU box:
#   tools/r03_p2_ablate.sh build ; gpurun -- 'bash tools/r03_p2_ablate.sh run'
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" == "build" ]; then
  cp "$ROOT/gparml_amd/libgparml_hip.so" "$ROOT/gparml_amd/lib_base.so.bin"
  for v in 1 2; do
    GPARML_OBJ_TAG=_abl$v GPARML_EXTRA_FLAGS="-DGPARML_ABLATE_P2_DMA=$v" GPARML_LIB_OUT="$ROOT/gparml_amd/lib_abl$v.so.bin" "$ROOT/tools/build_lib.sh"
  done
else
  bash "$ROOT/tools/ab_bench.sh" base abl1 abl2
fi
