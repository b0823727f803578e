#!/bin/bash
# A/B of single-mesh inference latency (tools/infer_trace.py N) between the default build and $1.
ALT="$1"; N="${2:-32}"; REPS="${3:-3}"
for i in $(seq "$REPS"); do
  for lib in "" "$ALT"; do
    ms=$(GEOBI_LIB="$lib" timeout -k 10 100 python tools/infer_trace.py "$N" 2>/dev/null | tail -1)
    echo "run $i  n=$N ${lib:-default}  $ms"
  done
done
