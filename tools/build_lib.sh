#!/bin/bash
# Build gparml_amd/libgparml_hip.so for gfx950 (in-tree).  Usage: tools/build_lib.sh [--asm]
# Objects are cached per source under build/obj (rebuilt when the source or any header is newer) and compiled in parallel.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/gparml_amd/csrc"
OUT="${GPARML_LIB_OUT:-$ROOT/gparml_amd/libgparml_hip.so}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC ${GPARML_EXTRA_FLAGS}"
if [ "$1" == "--asm" ]; then
  mkdir -p /tmp/asm
  for f in "$SRC"/*.hip; do
    b=$(basename "$f" .hip)
    (cd /tmp/asm && hipcc $FLAGS -c "$f" -save-temps=obj -o /tmp/asm/$b.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs:|Spill:|ScratchSize|Occupancy|LDS Size" || true)
  done
  exit 0
fi
# the object cache is keyed on the compile flags: an ablation / timing build (GPARML_EXTRA_FLAGS=-D...) never reuses production objects
FTAG=$(echo "$FLAGS" | md5sum | cut -c1-8)
OBJ="$ROOT/build/obj${GPARML_OBJ_TAG}-$FTAG"
mkdir -p "$OBJ"
newest_hdr=$(ls -t "$SRC"/*.h "$ROOT"/include/*.h | head -1)
pids=()
for f in "$SRC"/*.hip; do
  o="$OBJ/$(basename "$f" .hip).o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
    rm -f "$o"
    ( hipcc $FLAGS -c "$f" -o "$o" 2>&1 | grep -E "error|warning: v|Spill" || true; [ -f "$o" ] || { echo "FAILED: $f"; exit 1; } ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p || { echo "build failed"; exit 1; }; done
# link exactly the objects of the current sources (a deleted or renamed .hip leaves a stale object behind)
objs=()
for f in "$SRC"/*.hip; do objs+=("$OBJ/$(basename "$f" .hip).o"); done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
# the product library is stamped with the digest __graft_entry__.build() checks (sources + headers + its flags); variant builds are not
if [ "$OUT" == "$ROOT/gparml_amd/libgparml_hip.so" ] && [ -z "${GPARML_EXTRA_FLAGS}" ]; then
  python3 -c "import sys, glob, os; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; R = g.ROOT; print(g._source_digest(sorted(glob.glob(os.path.join(R, 'gparml_amd', 'csrc', '*.hip'))), glob.glob(os.path.join(R, 'gparml_amd', 'csrc', '*.h')) + glob.glob(os.path.join(R, 'include', '*.h'))))" > "$OUT.digest"
fi
ls -la "$OUT"
