# phase timelines (s_memtime stamps, diagnostic build) of the fused forward and the fused backward row pass in both tile
# geometries:  bash tools/build_variant.sh stamps feast_fused -DGEOBI_FUSED_STAMPS && bash tools/stamps_ab.sh
cd $GRAFT_REPO_ROOT
V=geobi_gnn_amd/csrc/build/variants/libgeobi_hip_stamps.so
for t in 1 0; do
  echo "=== GEOBI_TILE16=$t"
  GEOBI_TILE16=$t GEOBI_LIB=$V python tools/fused_stamps.py 64 32 2>/dev/null
  GEOBI_TILE16=$t GEOBI_LIB=$V python tools/k2_stamps.py 64 32 2>/dev/null
done
