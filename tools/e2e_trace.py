"""One mesh from raw points + faces to denoised vertices, repeated (for a rocprofv3 kernel trace of the preprocessing,
network and vertex-update kernels):  python tools/e2e_trace.py [n]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import network, meshgen, patches
dev = torch.device('cuda:0'); torch.manual_seed(0)
net = network.DualGNN().to(dev).eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=7)
pts = torch.from_numpy(noisy).to(dev); fv = torch.from_numpy(faces).to(dev).int(); gt = torch.from_numpy(clean).to(dev)
for _ in range(3): patches.predict_mesh(net, pts, fv, sub_size=10**9, gt_points=gt, patch_batch=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): patches.predict_mesh(net, pts, fv, sub_size=10**9, gt_points=gt, patch_batch=1)
torch.cuda.synchronize(); print('ms', (time.perf_counter() - t0) / 10 * 1e3)
