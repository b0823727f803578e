#!/bin/bash
# A/B builds of one translation unit with extra -D flags:  tools/build_variant.sh <name> <unit> <flags...>
#   -> geobi_gnn_amd/csrc/build/variants/libgeobi_hip_<name>.so   (use with GEOBI_LIB=...)
set -euo pipefail
NAME=$1; UNIT=$2; shift 2
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../geobi_gnn_amd/csrc" && pwd)"
mkdir -p "$HERE/build/variants"
# the other objects must match the tree the variant is compiled from: build.sh recompiles whatever is older than a header
# (round 3: a variant linked against stale objects is one of the two candidate causes of the kp_g1 fault)
bash "$HERE/build.sh" > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function "$@" -c "$HERE/$UNIT.hip" -o "$HERE/build/variants/${UNIT}_$NAME.o" 2>&1 | grep -E "error|Spill: [1-9]" || true
OBJS=()
for f in capi executor graph gemm feast feast_fused pool geom head_fused meshprep patch; do
  if [ "$f" = "$UNIT" ]; then OBJS+=("$HERE/build/variants/${UNIT}_$NAME.o"); else OBJS+=("$HERE/build/$f.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/build/variants/libgeobi_hip_$NAME.so" "${OBJS[@]}"
echo "built variant $NAME"
