"""Backward of the 128-channel FeaSt layers alone on the bench's level-1 / level-2 graph sizes (random regular graphs of
degree 13): us per backward.  GEOBI_ROWPASS_FUSED128=0 for the GEMM + standalone row pass."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import meshgen
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.feast_conv import FeaStConv
dev = torch.device('cuda:0')
pairs = [meshgen.synthetic_dual_data(16, (0.1, 0.2, 0.3)[i % 3], seed=200 + i) for i in range(4)]
dv, df = union_batch(pairs)
df = df.to(dev)
g = df.graph(df.x.shape[0]).ensure_in()
N = df.x.shape[0]
res = {'N': N}
for cin, cout in ((128, 128), (128, 64)):
    torch.manual_seed(0)
    conv = FeaStConv(cin, cout, 9).to(dev)
    x = torch.randn(N, cin, device=dev, requires_grad=True)
    gout = torch.randn(N, cout, device=dev)
    tot = 0.0
    for it in range(13):
        o = conv(x, g, slope=0.2)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); o.backward(gout); b.record(); torch.cuda.synchronize()
        if it >= 3:
            tot += a.elapsed_time(b)
    res['%d->%d bwd us' % (cin, cout)] = round(tot * 100, 1)
print(json.dumps(res))
