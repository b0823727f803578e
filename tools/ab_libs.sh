# same-box A/B of the product build against variant builds (tools/build_variant.sh):  bash tools/ab_libs.sh <variant> [<variant> ...]
cd $GRAFT_REPO_ROOT
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'])"; }
for v in "$@"; do
  GEOBI_LIB=geobi_gnn_amd/csrc/build/variants/libgeobi_hip_$v.so python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -m gpu -q -x -k "feast or properties" 2>&1 | tail -1
done
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | line "product   "
  for v in "$@"; do
    GEOBI_LIB=geobi_gnn_amd/csrc/build/variants/libgeobi_hip_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | line "$v"
  done
done
