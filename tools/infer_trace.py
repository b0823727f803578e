import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from geobi_gnn_amd import network, meshgen, infer
dev = torch.device('cuda:0'); torch.manual_seed(0)
net = network.DualGNN().to(dev).eval()
n = int(sys.argv[1])
dv, df = meshgen.synthetic_dual_data(n, 0.2, seed=7)
dv, df = dv.to(dev), df.to(dev)
for _ in range(3): infer.predict_one_submesh(net, (dv, df))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): infer.predict_one_submesh(net, (dv, df))
torch.cuda.synchronize(); print('ms', (time.perf_counter() - t0) / 10 * 1e3)
