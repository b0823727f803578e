#!/bin/bash
# A/B of one environment switch on ONE box: tools/ab_env.sh VAR [REPS] -> alternates VAR=1 / VAR=0 runs of the
# training bench and of single-mesh inference (n = 32).
VAR="$1"; REPS="${2:-3}"
for i in $(seq "$REPS"); do
  for v in 1 0; do
    ms=$(env "$VAR=$v" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null |
         python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    inf=$(env "$VAR=$v" timeout -k 10 100 python tools/infer_trace.py 32 2>/dev/null | tail -1)
    echo "run $i  $VAR=$v  train $ms ms/step   infer $inf"
  done
done
