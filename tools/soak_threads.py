"""Soak of the paths that run several library contexts at once: the grouped training step (bit-reproducibility over many
steps) and meshes in flight on worker threads (bit-equality with one-by-one), a few hundred rounds each.
    python tools/soak_threads.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobi_gnn_amd import network, meshgen, executor, patches
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.parallel import FlatParameters
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev)
bucket = FlatParameters(net).bucket
pairs = [meshgen.synthetic_dual_data(16, (0.1, 0.2, 0.3)[i % 3], seed=300 + i) for i in range(4)]
def union(ix):
    dv, df = union_batch([pairs[i] for i in ix])
    return dv.to(dev), df.to(dev)
tg = executor.TrainGroups(net, bucket).set_groups([union([0, 1]), union([2, 3])])
tg.step(); torch.cuda.synchronize()
ref = bucket.flat.clone(); bad = 0
t0 = time.time()
for r in range(rounds):
    tg.step(); torch.cuda.synchronize()
    bad += int(not torch.equal(bucket.flat, ref))
print('grouped step: %d rounds, %d differing buckets, %d sequential fallbacks, %.1f s' % (rounds, bad, tg.sequential_steps, time.time() - t0), flush=True)
net.eval()
meshes = []
for i, n in enumerate((12, 20, 16, 24, 14, 22, 18, 12)):
    noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=400 + i)
    meshes.append((torch.as_tensor(noisy, dtype=torch.float32, device=dev), torch.as_tensor(faces, dtype=torch.int32, device=dev), None))
want = [patches.predict_mesh(net, p, f, sub_size=3000, n_iter=10) for p, f, _ in meshes]
bad = 0; t0 = time.time()
for r in range(rounds // 4):
    for workers in (2, 3):
        got = patches.predict_many(net, meshes, workers=workers, sub_size=3000, n_iter=10)
        bad += sum(int(not (torch.equal(g['Np'], w['Np']) and torch.equal(g['V_updated'], w['V_updated']))) for g, w in zip(got, want))
print('meshes in flight: %d rounds x 2 worker counts x %d meshes, %d differing results, %.1f s' % (rounds // 4, len(meshes), bad, time.time() - t0), flush=True)
print('spin cap hits', executor.L.lib().geobi_net_spin_cap_hits())
