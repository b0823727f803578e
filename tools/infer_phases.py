"""Host vs device time of single-mesh inference (BASELINE configs[1])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobi_gnn_amd import network, meshgen, infer
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = network.DualGNN().to(dev).eval()
dv, df = meshgen.synthetic_dual_data(n, 0.2, seed=7)
dv, df = dv.to(dev), df.to(dev)
for _ in range(5): infer.predict_one_submesh(net, (dv, df))
torch.cuda.synchronize()
orig = torch.Tensor.tolist
wait = [0.0]
def timed(self):
    t0 = time.perf_counter(); r = orig(self); wait[0] += time.perf_counter() - t0; return r
torch.Tensor.tolist = timed
R = 50
t0 = time.perf_counter()
for _ in range(R): infer.predict_one_submesh(net, (dv, df))
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('n=%d: %.3f ms per forward; blocked in read-backs %.3f ms; final drain %.3f ms (total)' % (n, (t2 - t0) / R * 1e3, wait[0] / R * 1e3, (t2 - t1) * 1e3))
