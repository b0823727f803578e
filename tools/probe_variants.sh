# tools/probe_variants.sh <probe.py> <variant names...>: run a probe under the product build and each variant build
cd $GRAFT_REPO_ROOT
P=$1; shift
echo "== product"; python $P 2>&1 | tail -1
for v in "$@"; do echo "== $v"; GEOBI_LIB=geobi_gnn_amd/csrc/build/variants/libgeobi_hip_$v.so python $P 2>&1 | tail -1; done
