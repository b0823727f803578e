"""The chunked 128-channel backward kernel ALONE (no side-stream products beside it: GEOBI_OVERLAP=0) on the bench's level-1
graph sizes, for rocprofv3:   GEOBI_OVERLAP=0 rocprofv3 --kernel-trace --stats ... -- python3 tools/rp128_alone.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import meshgen
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.feast_conv import FeaStConv
dev = torch.device('cuda:0')
torch.manual_seed(0)
for freq, which in ((17, 'f'), (25, 'v')):          # facet graph of n = 17: 4 x 5 780 = 23 120 nodes; vertex graph of n = 25: 4 x 6 252
    pairs = [meshgen.synthetic_dual_data(freq, (0.1, 0.2, 0.3)[i % 3], seed=200 + i) for i in range(4)]
    dv, df = union_batch(pairs)
    d = (df if which == 'f' else dv).to(dev)
    g = d.graph(d.x.shape[0]).ensure_in()
    N = d.x.shape[0]
    conv = FeaStConv(128, 64, 9).to(dev)
    x = torch.randn(N, 128, device=dev, requires_grad=True)
    for _ in range(12):
        y = conv(x, g, slope=0.2)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    print(which, 'N', N, 'tiles of 32:', (N + 31) // 32, flush=True)
