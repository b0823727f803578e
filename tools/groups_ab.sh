# same-box comparison of the single-union step and the grouped step under a few settings (alternating)
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extra"
run() { echo -n "$1: "; shift; env "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
run "groups 1                 " $B --groups 1
run "groups 2 no side         " GEOBI_OVERLAP=0 $B --groups 2
run "groups 2 no side skew 250" GEOBI_OVERLAP=0 GEOBI_GROUP_SKEW_US=250 $B --groups 2
run "groups 2 no side skew 600" GEOBI_OVERLAP=0 GEOBI_GROUP_SKEW_US=600 $B --groups 2
run "groups 2 no side skew 1000" GEOBI_OVERLAP=0 GEOBI_GROUP_SKEW_US=1000 $B --groups 2
run "groups 4 no side         " GEOBI_OVERLAP=0 $B --groups 4
done
