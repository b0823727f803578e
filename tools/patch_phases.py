"""Where the time of a patch-split inference goes (n = 87 cut into patches of 20 000 faces): wall time per phase with a
device sync behind each (so the shares are upper bounds of what each phase costs when they overlap).
    python tools/patch_phases.py [freq] [sub_size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geobi_gnn_amd import network, meshgen, meshprep, patches, _lib as L
from geobi_gnn_amd.infer import predict_one_submesh

freq = int(sys.argv[1]) if len(sys.argv) > 1 else 87
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev).eval()
noisy, clean, faces = meshgen.noisy_icosphere(freq, 0.2, seed=7)
pts = torch.as_tensor(noisy, dtype=torch.float32, device=dev)
fv = torch.as_tensor(faces, dtype=torch.int32, device=dev)
for _ in range(2):
    patches.predict_mesh(net, pts, fv, sub_size=sub)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    r = patches.predict_mesh(net, pts, fv, sub_size=sub)
torch.cuda.synchronize()
print('predict_mesh n=%d sub_size=%d: %.2f ms end to end, %d patches' % (freq, sub, (time.perf_counter() - t0) / reps * 1e3, r['n_patches']))

stats = {}
for _ in range(3):
    patches.predict_mesh(net, pts, fv, sub_size=sub, stats=stats)
tot = sum(stats.values()) / 3
for k, v in stats.items():
    print('  %-45s %7.3f ms  (%4.1f %%)' % (k, v / 3 * 1e3, 100 * v / 3 / tot))
print('  %-45s %7.3f ms' % ('sum of the synchronised phases', tot * 1e3))
