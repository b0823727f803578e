"""Inference throughput on the BASELINE.json inference configs (not the driver's bench contract):
  configs[1]  Synthetic test_list single-mesh inference  -> icosphere n = 32 (F = 20 480)
  configs[3]  Kinect_Fusion large scan (~150 k faces)    -> icosphere n = 87 (F = 151 380), unsplit
Forward under no_grad + 60-sweep vertex update (test_dual.py:44-72), inputs resident in HBM."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import network, meshgen, infer, patches, _lib as L
import ctypes

dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev).eval()
out = []
for name, n, reps in (('configs[1] single mesh n=32', 32, 20), ('configs[3] large scan n=87', 87, 10)):
    dv, df = meshgen.synthetic_dual_data(n, 0.2, seed=7)
    meta = dv.meta
    edges = dv.edge_index.shape[1] + df.edge_index.shape[1]
    dv, df = dv.to(dev), df.to(dev)
    vf = meta['vf_indices'].to(dev)
    for _ in range(3):
        infer.predict_one(net, dv, df, meta['centroid'], meta['scale'], vf, 60)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        infer.predict_one_submesh(net, (dv, df))
    torch.cuda.synchronize(); t_net = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        infer.predict_one(net, dv, df, meta['centroid'], meta['scale'], vf, 60)
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / reps
    lib = L.lib(); lib.geobi_prof_enable(1)
    for _ in range(3):
        infer.predict_one_submesh(net, (dv, df))
    torch.cuda.synchronize()
    best = None
    for tag in (ci * 1000 + co for ci in (6, 12, 32, 64, 128) for co in (32, 64, 128)):
        k, ms, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        lib.geobi_prof_collect(tag, ctypes.byref(k), ctypes.byref(ms), ctypes.byref(by))
        if k.value and (best is None or ms.value > best[1]):
            best = ('%d->%d' % (tag // 1000, tag % 1000), ms.value, by.value, k.value)
    lib.geobi_prof_enable(0)
    out.append({'workload': name, 'faces': int(df.x.shape[0]), 'edges': edges,
                'network_ms': round(t_net * 1e3, 3), 'network_M_edges_per_s': round(edges / t_net / 1e6, 1),
                'with_vertex_update_ms': round(t_all * 1e3, 3),
                'fused_feast_kernel': {'layer': best[0], 'launches': best[3], 'avg_us': round(best[1] * 1e3 / best[3], 2),
                                     'algorithmic_GBps': round(best[2] / (best[1] * 1e-3) / 1e9, 1),
                                     'frac_of_8TBps': round(best[2] / (best[1] * 1e-3) / 1e9 / 8000, 3)}})
# end to end from the raw mesh (points + faces already in HBM): device preprocessing (graphs, normals,
# bilateral weights), optional patch split, network, merge, 60-sweep vertex update
for name, n, sub, reps, pb in (('configs[1] n=32 mesh -> denoised vertices', 32, 20480, 10, 4),
                               ('configs[3] n=87 mesh -> denoised vertices, unsplit (sub_size=200000)', 87, 200000, 5, 4),
                               ('configs[3] n=87 mesh -> denoised vertices, split at sub_size=20000, patches one by one', 87, 20000, 3, 1),
                               ('configs[3] n=87 mesh -> denoised vertices, split at sub_size=20000, 4 patches per pass', 87, 20000, 3, 4),
                               ('configs[3] n=87 mesh -> denoised vertices, split at sub_size=20000, 8 patches per pass', 87, 20000, 3, 8)):
    noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=7)
    pts = torch.from_numpy(noisy).to(dev)
    fv = torch.from_numpy(faces).to(dev).int()
    gt = torch.from_numpy(clean).to(dev)
    r = patches.predict_mesh(net, pts, fv, sub_size=sub, gt_points=gt, patch_batch=pb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r = patches.predict_mesh(net, pts, fv, sub_size=sub, patch_batch=pb)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
    out.append({'workload': name, 'faces': int(faces.shape[0]), 'patches': r['n_patches'],
                'end_to_end_ms': round(t * 1e3, 3), 'k_faces_per_s': round(faces.shape[0] / t / 1e3, 1)})
print(json.dumps(out, indent=1))
