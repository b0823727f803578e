"""Where the stand-in test list's time goes in patches.predict_batch: the small meshes and the patch-split meshes apart,
split groups of 1 / 2 / 4 meshes, patches per pass.   python tools/test_list_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobi_gnn_amd import network, meshgen, patches
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev).eval()
freqs, sigmas = (16, 22, 32, 45), (0.1, 0.2, 0.3)
meshes = []
for i in range(29):
    noisy, clean, faces = meshgen.noisy_icosphere(freqs[i % 4], sigmas[i % 3], seed=100 + i)
    meshes.append((torch.as_tensor(noisy, dtype=torch.float32, device=dev), torch.as_tensor(faces, dtype=torch.int32, device=dev),
                   torch.as_tensor(clean, dtype=torch.float32, device=dev)))
small = [m for m in meshes if m[1].shape[0] <= 20000]
big32 = [m for m in meshes if 20000 < m[1].shape[0] < 30000]
big45 = [m for m in meshes if m[1].shape[0] >= 30000]
def run(name, lst, **kw):
    with torch.no_grad():
        patches.predict_batch(net, lst, sub_size=20000, n_iter=60, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            patches.predict_batch(net, lst, sub_size=20000, n_iter=60, **kw)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('%-58s %7.2f ms  (%.2f ms per mesh)' % (name, dt * 1e3, dt * 1e3 / len(lst)), flush=True)
run('15 small meshes, unions <= 100 000 faces', small)
run('15 small meshes, one by one (max_faces 1)', small, max_faces=1)
for g in (1, 2, 4):
    run('7 meshes n = 32 (2 patches each), split groups of %d' % g, big32, split_group=g, patch_batch=8)
    run('7 meshes n = 45 (3 patches each), split groups of %d' % g, big45, split_group=g, patch_batch=8)
run('7 meshes n = 45, groups of 2, 5 patches per pass', big45, split_group=2, patch_batch=5)
run('7 meshes n = 45, groups of 4, 12 patches per pass', big45, split_group=4, patch_batch=12)
run('the whole list, groups of 2', meshes, split_group=2)
run('the whole list, groups of 4, 12 patches per pass', meshes, split_group=4, patch_batch=12)
