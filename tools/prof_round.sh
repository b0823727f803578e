# The measurement set of a round (run on the GPU box):  bash tools/prof_round.sh r02_final
# bench line, per-kernel rocprofv3 stats of the same command, PMC passes (FETCH_SIZE / WRITE_SIZE separately),
# inference configs, the training-driver run and the test-list evaluation.  Outputs under gpurun_out/<tag>_*.
set -e
TAG=${1:-r04_final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.err
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > /dev/null 2> gpurun_out/${TAG}_pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > /dev/null 2> gpurun_out/${TAG}_pmc_w.err
python tools/pmc_summary.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc_feast_fused.json > gpurun_out/${TAG}_pmc_summary.log 2>&1
# MFMA counters of every kernel that runs a node transform (own pass: --pmc never together with other tracing domains)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc/mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > /dev/null 2> gpurun_out/${TAG}_pmc_m.err
python tools/pmc_mfma.py gpurun_out/${TAG}_pmc/mfma gpurun_out/${TAG}_pmc_mfma.json > gpurun_out/${TAG}_pmc_mfma.log 2>&1
echo "pmc done"
python tools/bench_infer.py > gpurun_out/${TAG}_inference_configs.json 2> gpurun_out/${TAG}_infer.err
python bench.py --freq 16 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/${TAG}_bench_freq16.json 2>/dev/null
GEOBI_TILE16=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/${TAG}_bench_tile32.json 2>/dev/null
GEOBI_NET_EXECUTOR=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/${TAG}_bench_module_path.json 2>/dev/null
GEOBI_NET_EXECUTOR=0 GEOBI_FUSED=0 GEOBI_CHAIN_POOL=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/${TAG}_bench_round1_path.json 2>/dev/null
python bench.py --groups 2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/${TAG}_bench_groups2.json 2>/dev/null
python tools/patch_phases.py 87 20000 > gpurun_out/${TAG}_patch_phases.txt 2>&1
echo "inference + A/B lines done"
python tools/train_synthetic.py --max_epoch 40 --freq 32 --n_train 24 --n_eval 6 --lr 0.002 --lr_sch step --lr_step 12 --lr_decay 0.5 --batch_size 4 --out gpurun_out/${TAG}_net_freq32.pt > gpurun_out/${TAG}_train_synthetic_freq32.jsonl 2> gpurun_out/${TAG}_train.err
python tools/test_synthetic.py --model gpurun_out/${TAG}_net_freq32.pt --sub_size 20000 --json gpurun_out/${TAG}_test_synthetic.json > gpurun_out/${TAG}_test_synthetic.log 2>&1
echo "training + test list done"
cat gpurun_out/${TAG}_bench.json
