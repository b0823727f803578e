# Whole-step A/B of environment settings against the default, alternating processes, with the paired statistics
# (run on the GPU box):  bash tools/env_ab.sh PAIRS "NAME=V [NAME2=V2]" ["..." more settings]
# Every round runs the default and then each setting once; differences are per round (same minute on the same box).
PAIRS=$1; shift
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extra"
ms() { env "$@" $B 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for r in $(seq $PAIRS); do
  line="$(ms GEOBI_NOP=1)"
  for s in "$@"; do line="$line $(ms $s)"; done
  echo "$line"
done > gpurun_out/env_ab.txt
python - "$@" <<'PY'
import sys, statistics
names = ['default'] + sys.argv[1:]
rows = [[float(x) for x in l.split()] for l in open('gpurun_out/env_ab.txt') if l.strip()]
cols = list(zip(*rows))
print('%-44s %8s %8s   %s' % ('setting', 'mean ms', 'median', 'paired difference to the default (ms): mean +- standard error, rounds faster'))
for n, c in zip(names, cols):
    d = [x - y for x, y in zip(c, cols[0])]
    se = statistics.stdev(d) / len(d) ** 0.5 if len(d) > 1 else 0.0
    print('%-44s %8.3f %8.3f   %+.3f +- %.3f, %d of %d' % (n, statistics.mean(c), statistics.median(c), statistics.mean(d), se, sum(x < 0 for x in d), len(d)))
PY
