"""End-to-end time of a patch-split inference against the number of patches per network pass:
    python tools/patch_batch_sweep.py [freq] [sub_size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobi_gnn_amd import network, meshgen, patches
freq = int(sys.argv[1]) if len(sys.argv) > 1 else 87
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev).eval()
noisy, clean, faces = meshgen.noisy_icosphere(freq, 0.2, seed=7)
pts = torch.as_tensor(noisy, dtype=torch.float32, device=dev)
fv = torch.as_tensor(faces, dtype=torch.int32, device=dev)
for pb in (8, 1, 2, 3, 4, 5, 6, 8, 16):
    for _ in range(2):
        patches.predict_mesh(net, pts, fv, sub_size=sub, patch_batch=pb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        r = patches.predict_mesh(net, pts, fv, sub_size=sub, patch_batch=pb)
    torch.cuda.synchronize()
    print('patch_batch %2d: %.2f ms end to end (%d patches)' % (pb, (time.perf_counter() - t0) / 8 * 1e3, r['n_patches']), flush=True)
