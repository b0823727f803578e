"""Time the fused FeaSt kernels alone on the bench's level-0 graphs (4 meshes n = 32):
   python tools/fused_probe.py   [GEOBI_LIB=other build]  -> us per launch for the shapes of the network's big layers."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import meshgen, ops
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.feast_conv import FeaStConv

dev = torch.device('cuda:0')
pairs = [meshgen.synthetic_dual_data(32, (0.1, 0.2, 0.3)[i % 3], seed=200 + i) for i in range(4)]
dv, df = union_batch(pairs)
dv, df = dv.to(dev), df.to(dev)
res = {}
for name, d in (('facet', df), ('vertex', dv)):
    g = d.graph(d.x.shape[0]).ensure_in()
    N = d.x.shape[0]
    for cin, cout in ((64, 32), (32, 64), (12, 32), (128, 64)):
        torch.manual_seed(0)
        conv = FeaStConv(cin, cout, 9).to(dev)
        x = torch.randn(N, cin, device=dev, requires_grad=True)
        gout = torch.randn(N, cout, device=dev)
        for mode in ('fwd', 'fwd+bwd'):
            for _ in range(3):
                o = conv(x, g, slope=0.2)
                if mode != 'fwd':
                    o.backward(gout)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                o = conv(x, g, slope=0.2)
                if mode != 'fwd':
                    o.backward(gout)
            b.record(); torch.cuda.synchronize()
            res['%s N=%d %d->%d %s' % (name, N, cin, cout, mode)] = round(a.elapsed_time(b) * 100, 1)
print(json.dumps(res, indent=1))
