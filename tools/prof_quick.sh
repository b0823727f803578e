# quick state check on the GPU box: feast parity tests, bench line, per-kernel stats
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -m gpu -q -x -k "feast or properties" > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/${TAG}_tests.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.err
cat gpurun_out/${TAG}_bench.json
