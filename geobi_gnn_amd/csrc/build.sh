#!/bin/bash
# Builds libgeobi_hip.so for gfx950 (cross-compiles without a GPU).  Output stays in-tree so the
# .so travels to the GPU box with the repository snapshot.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libgeobi_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-result"
OBJS=()
mkdir -p "$HERE/build"
for f in capi executor graph gemm feast feast_fused pool geom head_fused meshprep patch; do
  src="$HERE/$f.hip"; obj="$HERE/build/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/common.h" -nt "$obj" ] || [ "$HERE/../../include/geobi_hip.h" -nt "$obj" ]; then
    echo "hipcc $f.hip"
    "$HIPCC" $FLAGS -c "$src" -o "$obj" &
  fi
  OBJS+=("$obj")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
