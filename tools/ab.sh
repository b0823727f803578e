#!/bin/bash
# A/B timing of two builds of libgeobi_hip.so on ONE box: alternates the default build with $1 (a second
# .so of the same ABI, e.g. geobi_gnn_amd/libgeobi_hip_A.so) REPS times and prints ms/step of each run.
#   tools/ab.sh geobi_gnn_amd/libgeobi_hip_A.so [REPS] [extra bench.py flags]
ALT="$1"; REPS="${2:-3}"; shift 2 || true
for i in $(seq "$REPS"); do
  for lib in "" "$ALT"; do
    ms=$(GEOBI_LIB="$lib" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null |
         python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "run $i  ${lib:-default}  $ms ms/step"
  done
done
