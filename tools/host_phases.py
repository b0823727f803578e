"""Wall time of the phases of one training step on a tiny mesh (GPU nearly idle -> host-side cost)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geobi_gnn_amd import network, net_util, ops
from geobi_gnn_amd.parallel import FlatParameters, batched_losses
dev = torch.device('cuda:0')
freq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net)
opt = torch.optim.Adam(flat.parameters(), lr=1e-3)
dv0, df0, edges = bench.make_batch(0, dev, freq)
for _ in range(5):
    bench.train_step(net, flat.bucket, opt, dv0, df0)
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    torch.cuda.synchronize()
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t
R = 30
for _ in range(R):
    t = time.perf_counter()
    flat.bucket.zero(); t = tick('zero', t)
    dv, df = dv0.shallow_copy(), df0.shallow_copy()
    x_v0 = dv.x
    feat_v = net.gnn_v(dv); t = tick('fwd gnn_v', t)
    verts = ops.HeadFn.apply(feat_v, net.fc_v1.weight, net.fc_v1.bias, net.fc_v2.weight, net.fc_v2.bias, 0, None, x_v0)
    fv32, cidx = network._fv_index(df, verts.shape[0])
    df.x = ops.FaceGeomFn.apply(verts, df.x, fv32, cidx); t = tick('fwd head_v+geom', t)
    feat_f = net.gnn_f(df); t = tick('fwd gnn_f', t)
    normals = ops.HeadFn.apply(feat_f, net.fc_f1.weight, net.fc_f1.bias, net.fc_f2.weight, net.fc_f2.bias, 1, None, None)
    lv, ln = batched_losses(verts, normals, dv0, df0); loss = network.dual_loss(lv, ln); t = tick('fwd head_f+loss', t)
    loss.backward(); t = tick('backward', t)
    opt.step(); t = tick('adam', t)
tot = sum(acc.values())
for k, v in acc.items():
    print('%-18s %7.3f ms' % (k, v / R * 1e3))
print('%-18s %7.3f ms (with a sync after every phase)' % ('total', tot / R * 1e3))
# inside one GNNModule forward: convs vs pooling
t = time.perf_counter(); acc = {}
for _ in range(R):
    dv = dv0.shallow_copy()
    t = time.perf_counter()
    g1 = dv.graph(dv.x.shape[0])
    x = net.gnn_v.l_conv1(dv.x, g1, slope=0.2); dv.x = x; t = tick('conv l1', t)
    d2 = net.gnn_v.pooling1(dv); t = tick('pooling1', t)
    g2 = d2.graph(); x2 = net.gnn_v.l_conv2(d2.x, g2, slope=0.2); d2.x = x2; t = tick('conv l2', t)
    d3 = net.gnn_v.pooling2(d2); t = tick('pooling2', t)
for k, v in acc.items():
    print('%-18s %7.3f ms' % (k, v / R * 1e3))
