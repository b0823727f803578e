# Per-kernel A/B of one environment switch under the kernel tracer (run on the GPU box):
#   bash tools/kernel_ab.sh NAME=VALUE_A NAME=VALUE_B [kernel-name substring] [pairs]
# Runs A, B, A, B ... as separate processes (some kernels have a fast and a slow mode per process: one pair proves nothing)
# and prints, per (kernel, grid), the average duration of every run and the per-step totals of the matching kernels.
set -e
export GEOBI_LIB_OLDER=1   # a library named through GEOBI_LIB may predate entry points of the header
A=$1; B=$2; PAT=${3:-feast_fused_kernel}; PAIRS=${4:-2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kab_*
i=0
for rep in $(seq $PAIRS); do
for setting in $A $B; do
  export $setting
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kab_$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/kab_$i.json 2> gpurun_out/kab_$i.err
  i=$((i+1))
done
done
python - "$A" "$B" "$PAT" $i <<'PY'
import csv, glob, collections, sys, re
A, B, pat, n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
res = []
for i in range(n):
    f = glob.glob(f'gpurun_out/kab_{i}/**/*kernel_trace.csv', recursive=True)[0]
    by = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].replace('void ', '').replace('geobi::(anonymous namespace)::', '').split('(')[0]
        if pat in k:
            k = re.sub(r', (1|2)>$', '>', k) if k.startswith('feast_fused_kernel') and k.count(',') >= 5 else k
            by[(k, int(row['Grid_Size_X']) // int(row['Workgroup_Size_X']))].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    res.append(by)
print('A = %s\nB = %s\nruns in order A B A B ...; us per launch' % (A, B))
print(f'{"kernel":48s} {"tiles":>6s} {"n":>4s} ' + ' '.join('%7s' % ('AB'[i % 2] + str(i // 2)) for i in range(n)))
tot = [0.0] * n
for key in sorted(res[0], key=lambda k: -sum(res[0][k])):
    if any(key not in r for r in res): continue
    ms = []
    for i, r in enumerate(res):
        v = r[key][len(r[key]) // 5:]
        ms.append(sum(v) / len(v)); tot[i] += sum(v) / 20.0 * 25 / 20
    print(f'{key[0]:48s} {key[1]:6d} {len(res[0][key]):4d} ' + ' '.join('%7.1f' % m for m in ms))
print('per step (us, matching kernels):'.ljust(61) + ' '.join('%7.1f' % t for t in tot))
PY
