# per-kernel comparison of the product build against variant builds on one box, alternating runs:
#   bash tools/kcmp.sh <name filter regex> <variant> [<variant> ...]    (tools/kprof.sh per run; prints avg us per kernel)
cd $GRAFT_REPO_ROOT
FILT=$1; shift
runs="- $* - $*"
i=0
for v in $runs; do
  i=$((i+1)); timeout -k 10 200 bash tools/kprof.sh cmp$i $v 80 > gpurun_out/kcmp_$i.txt 2>&1; echo "run $i ($v) done"
done
python - "$FILT" $runs <<'PY'
import re, sys
filt = re.compile(sys.argv[1]); names = sys.argv[2:]
tabs = []
for i in range(len(names)):
    d = {}
    for l in open('gpurun_out/kcmp_%d.txt' % (i + 1)):
        m = re.match(r"(\S+(?:, \S+)*)\s+calls/step\s+([\d.]+)\s+avg\s+([\d.]+) us\s+per step\s+([\d.]+)", l)
        if m and filt.search(m.group(1)): d[m.group(1)] = (float(m.group(3)), float(m.group(4)))
    tabs.append(d)
print('%-50s' % 'kernel' + ''.join('%10s' % n for n in names))
tot = [0.0] * len(names)
for k in tabs[0]:
    if all(k in t for t in tabs):
        print('%-50s' % k[:50] + ''.join('%10.1f' % t[k][0] for t in tabs))
        for i, t in enumerate(tabs): tot[i] += t[k][1]
print('%-50s' % 'per step (us)' + ''.join('%10.0f' % x for x in tot))
PY
