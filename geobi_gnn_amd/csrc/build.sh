#!/bin/bash
# Builds libgeobi_hip.so for gfx950 (cross-compiles without a GPU).  Output stays in-tree so the
# .so travels to the GPU box with the repository snapshot.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libgeobi_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-result"
OBJS=()
PIDS=()
mkdir -p "$HERE/build"
# every header is a dependency of every object (feast_dev.h is shared by feast.hip and feast_fused.hip; a stale
# object would ship in the in-tree .so), and so is this script (compiler flags)
HDRS=("$HERE"/*.h "$HERE/../../include"/*.h "$HERE/build.sh")
for f in capi executor graph gemm feast feast_fused pool geom head_fused meshprep patch; do
  src="$HERE/$f.hip"; obj="$HERE/build/$f.o"
  stale=0
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then stale=1; fi
  for h in "${HDRS[@]}"; do if [ "$h" -nt "$obj" ]; then stale=1; fi; done
  if [ $stale = 1 ]; then
    echo "hipcc $f.hip"
    rm -f "$obj"                      # a failed compile must not leave the previous object to be linked
    "$HIPCC" $FLAGS -c "$src" -o "$obj" &
    PIDS+=($!)
  fi
  OBJS+=("$obj")
done
for p in "${PIDS[@]}"; do wait "$p"; done      # set -e: the first failed compile stops the build
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
