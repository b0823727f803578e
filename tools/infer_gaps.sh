# Kernel time against wall time of single-mesh inference (run on the GPU box): bash tools/infer_gaps.sh [n]
set -e
N=${1:-32}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/infer_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/infer_trace -- python3 tools/infer_trace.py $N > gpurun_out/infer_trace.log 2>&1
python tools/infer_trace.py $N
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/infer_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('void ', '').replace('geobi::(anonymous namespace)::', '').split('(')[0][:44] for r in rows]
# one forward = from one head_fwd... find period by the last kernel of a forward (head_fwd_fused_kernel<3> appears twice per forward? use count)
per = len(rows) // 13
last = rows[-per:]
ln = names[-per:]
wall = (int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e3
gaps = [(int(last[i + 1]['Start_Timestamp']) - int(last[i]['End_Timestamp'])) / 1e3 for i in range(len(last) - 1)]
print('launches per forward %d; traced wall %.0f us, kernels %.0f us, gaps %.0f us (median gap %.1f us)' % (per, wall, busy, sum(g for g in gaps if g > 0), sorted(gaps)[len(gaps) // 2]))
big = sorted([(g, ln[i], ln[i + 1]) for i, g in enumerate(gaps)], reverse=True)[:8]
for g, a, b in big: print('  gap %6.1f us  after %-44s before %s' % (g, a, b))
tot = collections.defaultdict(float); cnt = collections.Counter()
for r, n in zip(last, ln):
    tot[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; cnt[n] += 1
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:14]: print('  %-46s %3d launches %7.1f us' % (n, cnt[n], v))
PY
