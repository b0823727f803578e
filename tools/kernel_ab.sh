# Per-kernel A/B of one environment switch under the kernel tracer (run on the GPU box):
#   bash tools/kernel_ab.sh NAME=VALUE_A NAME=VALUE_B [kernel-name substring]
# prints, per (kernel, grid), the average duration in both settings and the per-step totals of the matching kernels.
set -e
A=$1; B=$2; PAT=${3:-feast_fused_kernel}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for setting in $A $B; do
  export $setting
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kab_$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/kab_$i.json 2> gpurun_out/kab_$i.err
  i=$((i+1))
done
python - "$A" "$B" "$PAT" <<'PY'
import csv, glob, collections, sys, re
A, B, pat = sys.argv[1:4]
res = []
for i in (0, 1):
    f = glob.glob(f'gpurun_out/kab_{i}/**/*kernel_trace.csv', recursive=True)[0]
    by = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].replace('void ', '').replace('geobi::(anonymous namespace)::', '').split('(')[0]
        if pat in k:
            k = re.sub(r', (1|2)>$', '>', k) if k.count(',') >= 5 else k      # fold the column-part instantiations
            by[(k, int(row['Grid_Size_X']) // int(row['Workgroup_Size_X']))].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    res.append(by)
print(f'{"kernel":48s} {"tiles":>6s} {"n":>4s} {A:>26s} {B:>26s}')
ta = tb = 0.0
for key in sorted(res[0], key=lambda k: -sum(res[0][k])):
    a = res[0][key]; b = res[1].get(key, [])
    a = a[len(a) // 5:]; b = b[len(b) // 5:]
    if not b: continue
    ma, mb = sum(a) / len(a), sum(b) / len(b)
    ta += sum(a) / 20.0; tb += sum(b) / 20.0
    print(f'{key[0]:48s} {key[1]:6d} {len(a):4d} {ma:26.1f} {mb:26.1f}')
print('per step (us, matching kernels): %.1f  %.1f' % (ta * 25 / 20, tb * 25 / 20))
PY
