"""Where the fresh-batch step's extra time goes (bench.py extra.fresh_batch): union, per-union structures, step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geobi_gnn_amd import network, meshgen, meshprep
from geobi_gnn_amd.data import union_batch_graphs
from geobi_gnn_amd.parallel import FlatParameters
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net); bucket = flat.bucket
opt = torch.optim.Adam(flat.parameters(), lr=1e-3, fused=True)
raw = [meshgen.noisy_icosphere(32, (0.1, 0.2, 0.3)[i % 3], seed=500 + i) for i in range(12)]
from geobi_gnn_amd.network import _fv_index
pool = [meshprep.build_dual_data(n, f, c, device=dev) for n, c, f in raw]
if os.environ.get('PREPARED', '1') != '0':       # per-mesh structures built once per mesh: the union carries them over
    for dv_, df_ in pool:
        dv_.graph().ensure_in(); df_.graph().ensure_in()
        _fv_index(df_, dv_.x.shape[0])[1].get()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
def step(k, parts):
    idx = [(k + 3 * i) % 12 for i in range(4)]
    t0 = sync()
    dv, df = union_batch_graphs([pool[i] for i in idx])
    t1 = sync()
    gv, gf = dv.graph().ensure_in(), df.graph().ensure_in()
    t2 = sync()
    from geobi_gnn_amd.network import _fv_index
    _, corner = _fv_index(df, dv.x.shape[0]); corner.get()
    t3 = sync()
    from geobi_gnn_amd.parallel import _mesh_weights
    _mesh_weights(dv); _mesh_weights(df)
    t4 = sync()
    bench.train_step(net, bucket, opt, dv, df, collective=False)
    t5 = sync()
    parts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
parts = []
for k in range(4): step(k, [])
for k in range(12): step(4 + k, parts)
import numpy as np
print('per-step ms of the step phase:', [round(p[4] * 1e3, 2) for p in parts])
m = np.array(parts).mean(0) * 1e3
print('ms: union %.3f | reverse-edge index %.3f | corner lists %.3f | loss weights %.3f | step on prepared batch %.3f | sum %.3f' % (*m, m.sum()))
