cd $GRAFT_REPO_ROOT
for v in unr2 unr4 unr8; do echo "== $v"; GEOBI_LIB=geobi_gnn_amd/csrc/build/variants/libgeobi_hip_$v.so python tools/fused_probe.py 2>&1 | grep -v "^{\|^}" ; done > gpurun_out/probe1.log 2>&1
echo "== unfused" >> gpurun_out/probe1.log; GEOBI_FUSED=0 python tools/fused_probe.py 2>&1 | grep -v "^{\|^}" >> gpurun_out/probe1.log
