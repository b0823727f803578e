# per-kernel rocprofv3 stats of a short bench run, product build or a variant:  bash tools/kprof.sh <tag> [variant] [top]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; V=$2; TOP=${3:-40}
if [ -n "$V" ] && [ "$V" != "-" ]; then export GEOBI_LIB=geobi_gnn_amd/csrc/build/variants/libgeobi_hip_$V.so; fi
rm -rf gpurun_out/kp_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp_$TAG -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > /dev/null 2> gpurun_out/kp_$TAG.err
python tools/kstats.py gpurun_out/kp_$TAG 13 $TOP
