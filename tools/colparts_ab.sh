# Same-box A/B of the fused kernel's column parts (run on the GPU box): bash tools/colparts_ab.sh
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extra"
run() { echo -n "$1: "; shift; env "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
run "parts 1 (off)        " GEOBI_COLUMN_PARTS=1 $B
run "per launch, <= 1024  " GEOBI_COLUMN_PARTS=0 $B
run "per launch, <= 512   " GEOBI_COLUMN_PARTS=0 GEOBI_COLUMN_PARTS_MAX_TILES=512 $B
run "per launch, <= 1536  " GEOBI_COLUMN_PARTS=0 GEOBI_COLUMN_PARTS_MAX_TILES=1536 $B
done
