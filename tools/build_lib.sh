#!/bin/bash
# Build gparml_amd/libgparml_hip.so for gfx950 (in-tree). Usage: tools/build_lib.sh [--asm]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/gparml_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
if [ "$1" == "--asm" ]; then
  mkdir -p /tmp/asm
  for f in "$SRC"/*.hip; do
    b=$(basename "$f" .hip)
    (cd /tmp/asm && hipcc $FLAGS -c "$f" -save-temps=obj -o /tmp/asm/$b.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs:|Spill:|ScratchSize|Occupancy|LDS Size" || true)
  done
else
  hipcc $FLAGS -shared "$SRC"/*.hip -o "$ROOT/gparml_amd/libgparml_hip.so" 2>&1 | grep -E "error|warning: v|Spill" || true
  ls -la "$ROOT/gparml_amd/libgparml_hip.so"
fi
