set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 > gpurun_out/r02_v1_bench.json 2> gpurun_out/r02_v1_bench.err
GEOBI_FUSED=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r02_v1_bench_unfused.json 2> gpurun_out/r02_v1_bench_unfused.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r02_v1_bench_under_rocprof.json 2> gpurun_out/r02_prof1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_pmc/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> gpurun_out/r02_pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_pmc/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> gpurun_out/r02_pmc_w.err
python tools/pmc_summary.py gpurun_out/r02_pmc gpurun_out/r02_pmc_feast_fused.json > gpurun_out/r02_pmc_summary.log 2>&1
cat gpurun_out/r02_v1_bench.json; cat gpurun_out/r02_v1_bench_unfused.json
