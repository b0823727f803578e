# same-box A/B of the product build against a variant build:  tools/ab_lib.sh <variant name> [test -k filter]
cd $GRAFT_REPO_ROOT
V=geobi_gnn_amd/csrc/build/variants/libgeobi_hip_$1.so
GEOBI_LIB=$V python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -m gpu -q -x -k "${2:-feast or properties}" 2>&1 | tail -2
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('product', d['value'], d['ms_per_step'])"
  GEOBI_LIB=$V python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('variant', d['value'], d['ms_per_step'])"
done
python tools/k2_probe.py; GEOBI_LIB=$V python tools/k2_probe.py
