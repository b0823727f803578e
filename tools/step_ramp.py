"""Per-step wall time of the bench's training step from process start (is the first process on a fresh box slower, and
for how long?):  python tools/step_ramp.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geobi_gnn_amd import network
from geobi_gnn_amd.parallel import FlatParameters
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net); bucket = flat.bucket
opt = torch.optim.Adam(flat.parameters(), lr=1e-3, fused=True)
dv, df, edges = bench.make_batch(0, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ts = []
for it in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.train_step(net, bucket, opt, dv, df, collective=False)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('ms per step:', ' '.join('%.1f' % t for t in ts))
