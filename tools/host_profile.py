"""cProfile of the host side of one training step (run on the GPU box)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geobi_gnn_amd import network
from geobi_gnn_amd.parallel import FlatParameters
dev = torch.device('cuda:0')
freq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net)
from geobi_gnn_amd.train_util import FlatAdam
opt = FlatAdam(flat.parameters(), lr=1e-3)
dv, df, edges, _ = bench.make_batch(0, dev, freq)
for _ in range(5):
    bench.train_step(net, flat.bucket, opt, dv, df)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    bench.train_step(net, flat.bucket, opt, dv, df)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('tottime')
ps.print_stats(45)
print(s.getvalue())
