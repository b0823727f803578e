"""cProfile of the host side of single-mesh inference (which Python / ctypes calls the 2.4 ms go to)."""
import cProfile, pstats, sys, os, io
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
pr = cProfile.Profile()
pr.enable()
for _ in range(50): infer.predict_one_submesh(net, (dv, df))
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue())
