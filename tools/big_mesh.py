"""Largest-size sanity run: one mesh of ~0.5-0.8 M faces through device preprocessing, the network
(forward + backward) and the vertex update; checks finiteness, unit normals and bitwise reproducibility."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import network, meshgen, meshprep, patches

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=1)
pts, gt = torch.from_numpy(noisy).to(dev), torch.from_numpy(clean).to(dev)
fv = torch.from_numpy(faces).to(dev).int()
torch.manual_seed(0)
net = network.DualGNN().to(dev)
t0 = time.time()
dv, df = meshprep.build_dual_data(pts, fv, gt, device=dev)
torch.cuda.synchronize(); t_prep = time.time() - t0
outs = []
for rep in range(3):            # the first two calls size (and re-size) the arena
    net.zero_grad(set_to_none=True)
    t0 = time.time()
    vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy()))
    loss = network.dual_loss(network.loss_v(vp, dv.y, 'L1'), network.loss_n(npred, df.y, 'L1'))
    loss.backward()
    torch.cuda.synchronize(); t_step = time.time() - t0
    outs.append((vp.detach().clone(), npred.detach().clone(), net.gnn_f.r_conv4.lin.weight.grad.clone()))
same = all(torch.equal(a, b) for a, b in zip(outs[1], outs[2]))
net.eval()
r = patches.predict_mesh(net, pts, fv, sub_size=10 ** 9, gt_points=gt)
edges = dv.graph().E + df.graph().E + dv.x.shape[0] + df.x.shape[0]
print(json.dumps({'n': n, 'faces': int(fv.shape[0]), 'vertices': int(pts.shape[0]), 'edges_incl_loops': int(edges),
                  'prep_ms': round(t_prep * 1e3, 2), 'fwd_bwd_ms': round(t_step * 1e3, 2),
                  'M_edges_per_s_fwd_bwd': round(edges / t_step / 1e6, 1), 'loss': float(loss),
                  'finite': bool(torch.isfinite(vp).all() and torch.isfinite(npred).all()),
                  'unit_normals_max_dev': float((npred.detach().norm(dim=1) - 1).abs().max()),
                  'bitwise_reproducible': bool(same), 'angle1_deg': r['angle1'],
                  'max_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
from geobi_gnn_amd import executor
print('executor', executor.STATS)
