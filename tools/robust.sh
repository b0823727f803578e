cd $GRAFT_REPO_ROOT
python tools/size_sweep.py > gpurun_out/r03_size_sweep.log 2>&1; echo "size_sweep rc=$?"
tail -4 gpurun_out/r03_size_sweep.log
timeout -k 10 300 python tools/big_mesh.py 160 > gpurun_out/r03_big_mesh.json 2> gpurun_out/r03_big_mesh.err; echo "big_mesh rc=$?"; cat gpurun_out/r03_big_mesh.json
python -c "
import sys; sys.path.insert(0,'.')
from geobi_gnn_amd import executor
print('executor stats', executor.STATS)"
python -m pytest tests/test_gpu_bench.py -m gpu -q 2>&1 | tail -2
