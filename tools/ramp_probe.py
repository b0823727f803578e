"""Does the step time of a process settle?  The bench's step, timed in blocks of 10 from the first step of a fresh process.
   python tools/ramp_probe.py [blocks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from geobi_gnn_amd import network
from geobi_gnn_amd.parallel import FlatParameters
from geobi_gnn_amd.train_util import FlatAdam
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net)
opt = FlatAdam(flat.parameters(), lr=1e-3)
dv, df, edges, _ = bench.make_batch(0, dev, 32)
out = []
for b in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        bench.train_step(net, flat.bucket, opt, dv, df)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) * 100, 3))
print('ms per step, blocks of 10 steps:', out)
