"""How much of a step the host spends blocked in the pooling size read-backs (tolist) and in the final sync."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geobi_gnn_amd import network
from geobi_gnn_amd.parallel import FlatParameters
dev = torch.device('cuda:0')
freq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net)
opt = torch.optim.Adam(flat.parameters(), lr=1e-3, fused=True)
dv, df, edges = bench.make_batch(0, dev, freq)
for _ in range(5):
    bench.train_step(net, flat.bucket, opt, dv, df)
torch.cuda.synchronize()
from geobi_gnn_amd import _lib as L
orig = L.read_i32
wait = [0.0, 0]
def timed(t, n=None):
    t0 = time.perf_counter(); r = orig(t, n); wait[0] += time.perf_counter() - t0; wait[1] += 1; return r
L.read_i32 = timed
R = 30
t0 = time.perf_counter()
for _ in range(R):
    bench.train_step(net, flat.bucket, opt, dv, df)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('freq %d: step %.3f ms; host blocked in %d read-backs/step: %.3f ms/step; final drain %.3f ms total' %
      (freq, (t2 - t0) / R * 1e3, wait[1] // R, wait[0] / R * 1e3, (t2 - t1) * 1e3))
