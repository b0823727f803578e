"""Host-side time of one training step's phases (wall clock of the Python calls, no synchronisation inside a step):
where the CPU spends its time while the GPU runs.   python tools/host_step.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geobi_gnn_amd import network, meshgen
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.parallel import FlatParameters, batched_losses

dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net)
bucket = flat.bucket
opt = torch.optim.Adam(flat.parameters(), lr=1e-3, fused=True)
pairs = [meshgen.synthetic_dual_data(32, (0.1, 0.2, 0.3)[i % 3], seed=100 + i) for i in range(4)]
dv, df = union_batch(pairs); dv, df = dv.to(dev), df.to(dev)
acc = {}
def mark(k, t0):
    t = time.perf_counter(); acc[k] = acc.get(k, 0.0) + (t - t0); return t
for it in range(25):
    if it == 5:
        torch.cuda.synchronize(); acc.clear(); T0 = time.perf_counter()
    t = time.perf_counter()
    bucket.zero(); t = mark('zero', t)
    vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy())); t = mark('forward call', t)
    lv, ln = batched_losses(vp, npred, dv, df, 'L1', 'L1'); loss = network.dual_loss(lv, ln); t = mark('loss', t)
    loss.backward(); t = mark('backward call', t)
    opt.step(); t = mark('optimizer', t)
torch.cuda.synchronize(); T1 = time.perf_counter()
n = 20
print('step wall %.3f ms' % ((T1 - T0) / n * 1e3))
for k, v in acc.items(): print('  host %-14s %.3f ms' % (k, v / n * 1e3))
print('  host total      %.3f ms' % (sum(acc.values()) / n * 1e3))
from geobi_gnn_amd import executor
print('executor', executor.STATS, 'enabled', executor.ENABLED)
