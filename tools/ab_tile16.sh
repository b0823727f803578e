# same-box A/B of the fused kernels' tile geometry (GEOBI_TILE16=1: 16-row tiles, four workgroups per CU; 0: 32-row):
# parity tests under both, the layer probes, then alternating bench lines.   bash tools/ab_tile16.sh [tag]
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-ab16}
for v in 1 0; do
  GEOBI_TILE16=$v python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -m gpu -q -x -k "feast or properties" > gpurun_out/${TAG}_tests_$v.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests_$v.log; exit 1; }
  echo "GEOBI_TILE16=$v: $(tail -1 gpurun_out/${TAG}_tests_$v.log)"
done
for v in 1 0; do
  echo "--- fused_probe GEOBI_TILE16=$v"; GEOBI_TILE16=$v python tools/fused_probe.py
  echo "--- k2_probe GEOBI_TILE16=$v"; GEOBI_TILE16=$v python tools/k2_probe.py
done
for i in 1 2 3; do
  for v in 1 0; do
    GEOBI_TILE16=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('GEOBI_TILE16=$v', d['value'], d['ms_per_step'])"
  done
done
