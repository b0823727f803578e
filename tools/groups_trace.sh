# kernel trace of bench.py with G mesh groups: bash tools/groups_trace.sh <G> <tag> [extra env assignments...]
set -e
G=$1; TAG=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --groups $G --steps 10 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_prof.err
python tools/trace_groups.py gpurun_out/${TAG}_prof 6 > gpurun_out/${TAG}_trace.txt 2>&1 || true
cat gpurun_out/${TAG}_bench.json | head -c 400; echo; cat gpurun_out/${TAG}_trace.txt
