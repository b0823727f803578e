# Same-box alternating A/B of two builds of the library, whole step, untraced (run on the GPU box):
#   bash tools/lib_ab.sh geobi_gnn_amd/csrc/build/variants/libgeobi_hip_base.so [pairs]
BASE=$PWD/$1; PAIRS=${2:-4}
export GEOBI_LIB_OLDER=1
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extra"
run() { echo -n "$1: "; shift; env "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in $(seq $PAIRS); do
run "base" GEOBI_LIB=$BASE $B
run "new " GEOBI_LIB= $B
done
