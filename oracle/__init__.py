"""TEST INFRASTRUCTURE -- CPU oracle for the bi-domain mesh-graph convolution path.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import
anything under oracle/; the product (geobi_gnn_amd) never does and fails loudly when its
HIP library is missing.

Parity status
-------------
* Orchestration (GNNModule / DualGNN / PoolingLayer / pool helpers / losses /
  computer_face_normal / calc_weight / update_position2): restated in ref_model.py and
  checked bit-for-bit against the reference's own files imported in the build container
  (oracle/gen_golden.py, fixtures in tests/golden/).
* Third-party primitives (FeaStConv, scatter, coalesce, graclus, consecutive_cluster):
  PARITY UNPINNED -- the wheels are absent, un-pinned, and the reference ships no test,
  golden vector or checkpoint for them.  pyg_ops.py restates their
  published algorithms.
"""
