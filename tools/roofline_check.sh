cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print({k: r[k] for k in ('frac','avg_us','launches')}, r['all_instantiations'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rf_prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > /dev/null 2> gpurun_out/rf.err
grep "feast_fused_kernel<64, 0, 0, 1, 16" $(ls -t gpurun_out/rf_prof/*/*kernel_stats.csv | head -1) | awk -F'",' '{print "rocprofv3:", $2}'
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print({k: r[k] for k in ('frac','avg_us','launches')}, r['all_instantiations'])"
