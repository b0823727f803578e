"""Same-box A/B of the whole-network executor against the module-by-module inference path."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import network, meshgen, infer, patches, executor
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev).eval()
res = {}
def timeit(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n, reps in ((16, 40), (32, 40), (87, 10)):
    dv, df = meshgen.synthetic_dual_data(n, 0.2, seed=7)
    dv, df = dv.to(dev), df.to(dev)
    for rnd in range(2):
        for en in (True, False):
            executor.ENABLED = en
            res['n=%d network %s #%d' % (n, 'executor' if en else 'modules ', rnd)] = round(timeit(lambda: infer.predict_one_submesh(net, (dv, df)), reps), 3)
noisy, clean, faces = meshgen.noisy_icosphere(87, 0.2, seed=7)
pts = torch.from_numpy(noisy).to(dev); fv = torch.from_numpy(faces).to(dev).int()
for pb in (1, 8):
    for rnd in range(2):
        for en in (True, False):
            executor.ENABLED = en
            res['n=87 split 20000, %d per pass %s #%d' % (pb, 'executor' if en else 'modules ', rnd)] = round(timeit(lambda: patches.predict_mesh(net, pts, fv, sub_size=20000, patch_batch=pb), 3), 3)
print(json.dumps(res, indent=1))
print('executor stats', executor.STATS)
