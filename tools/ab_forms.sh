# same-box A/B of the 64-channel backward row-pass forms: the full GPU suite, then alternating bench runs
#   (staged, chunked64) = default | (0,0) lane-private | (1,0) staged everywhere | (1,1) chunked everywhere
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
run() { GEOBI_ROWPASS_STAGED=$1 GEOBI_ROWPASS_CHUNKED64=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('staged=$1 chunked64=$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('default             ', d['value'], d['ms_per_step'])"
  run 0 0; run 1 0
done
