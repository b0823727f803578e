# rocprofv3 per-kernel averages of tools/k2_probe.py (single-layer backward on the bench's facet graph) under variant builds:
#   bash tools/kprobe.sh <name filter> <variant|-> [<variant> ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FILT=$1; shift
for V in "$@"; do
  if [ "$V" != "-" ]; then export GEOBI_LIB=geobi_gnn_amd/csrc/build/variants/libgeobi_hip_$V.so; else unset GEOBI_LIB; fi
  rm -rf gpurun_out/kpr_$V
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kpr_$V -- python3 tools/k2_probe.py > /dev/null 2> gpurun_out/kpr_$V.err
  echo "== $V"
  python - "$V" "$FILT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob('gpurun_out/kpr_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r['Name']:
        nm = r['Name'].replace('void geobi::(anonymous namespace)::', '').split('(')[0]
        print('  %-60s calls %4s avg %8.1f us' % (nm, r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
