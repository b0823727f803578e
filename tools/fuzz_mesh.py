"""Randomised sweep, third part: IRREGULAR meshes -- noisy icospheres with a random share of their faces removed (holes,
boundaries, vertices without faces, several connected components).  Device preprocessing (graphs, incidence, normals,
bilateral weights, features) against the host generator; the device patch split against the sequential statement (a patch
stops early when its component is exhausted); the whole inference of such a mesh with and without the patch split running
through (finite, unit normals).      python tools/fuzz_mesh.py [seconds] [seed]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_gpu_meshprep as TM
import test_gpu_patches as TP
from geobi_gnn_amd import meshgen, patches, network
from oracle import mesh_ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev).eval()
count = {'preprocessing': 0, 'split': 0, 'inference': 0}
fails = []
t_end = time.time() + budget
while time.time() < t_end:
    n = rng.choice([2, 3, 5, 8, 12, 16]); drop = rng.choice([0.0, 0.05, 0.15, 0.3, 0.5]); seed = rng.randrange(100000)
    noisy, clean, faces = meshgen.noisy_icosphere(n, rng.choice([0.1, 0.3]), seed)
    keep = np.random.default_rng(seed).random(faces.shape[0]) >= drop
    faces = np.ascontiguousarray(faces[keep])
    if faces.shape[0] < 4:
        continue
    tag = (n, drop, seed, int(faces.shape[0]))
    try:
        TM._check_against_meshgen(dev, noisy, faces, clean)
    except AssertionError as ex:
        fails.append(('preprocessing', tag, str(ex)[:200]))
    except Exception as ex:                                  # noqa: BLE001
        fails.append(('preprocessing raised', tag, repr(ex)[:200]))
    count['preprocessing'] += 1
    pts = torch.as_tensor(noisy, dtype=torch.float32, device=dev); fv = torch.as_tensor(faces, dtype=torch.int32, device=dev)
    sub = rng.choice([20, 100, 700, 3000])
    try:
        got = [s.cpu().numpy() for s in patches.split_faces(pts, fv, sub)]
        rp, ls = TP._host_incidence(faces, pts.shape[0])
        d2 = ((pts[fv.long()].mean(1) - pts.mean(0, keepdim=True)) ** 2).sum(1).cpu().numpy()
        want = mesh_ops.split_faces(d2, np.ascontiguousarray(faces, dtype=np.int32), rp, ls, sub)
        if not (len(got) == len(want) and all(np.array_equal(a, b) for a, (_, b) in zip(got, want))):
            fails.append(('split', tag, sub, len(got), len(want)))
    except Exception as ex:                                  # noqa: BLE001
        fails.append(('split raised', tag, sub, repr(ex)[:200]))
    count['split'] += 1
    try:
        with torch.no_grad():
            whole = patches.predict_mesh(net, pts, fv, sub_size=10 ** 9, n_iter=5)
            cut = patches.predict_mesh(net, pts, fv, sub_size=max(sub, 50), n_iter=5)
        used = torch.zeros(pts.shape[0], dtype=torch.bool, device=dev)
        used[fv.long().reshape(-1)] = True
        for r in (whole, cut):
            # a vertex no face uses is 0 / 0 after the overlap merge, as in the reference (test_dual.py:56); checked: the rest
            m = used if r['n_patches'] > 1 else torch.ones_like(used)
            ok = bool(torch.isfinite(r['Vp'][m]).all() and torch.isfinite(r['V_updated'][m]).all() and torch.isfinite(r['Np']).all())
            nrm = r['Np'].norm(dim=1)
            ok = ok and bool(((nrm - 1).abs() < 1e-4).all())
            if not ok:
                fails.append(('inference', tag, sub, r['n_patches']))
    except Exception as ex:                                  # noqa: BLE001
        fails.append(('inference raised', tag, sub, repr(ex)[:300]))
    count['inference'] += 1
# ---- lists of irregular meshes through predict_batch against mesh-by-mesh inference, bit for bit
t_end = time.time() + budget / 3
count['batch'] = 0
while time.time() < t_end:
    lst = []
    for _ in range(rng.choice([2, 3, 5])):
        n = rng.choice([3, 5, 8, 12]); drop = rng.choice([0.0, 0.1, 0.3]); seed = rng.randrange(100000)
        noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed)
        faces = np.ascontiguousarray(faces[np.random.default_rng(seed).random(faces.shape[0]) >= drop])
        if faces.shape[0] >= 4:
            lst.append((torch.as_tensor(noisy, dtype=torch.float32, device=dev), torch.as_tensor(faces, dtype=torch.int32, device=dev),
                        torch.as_tensor(clean, dtype=torch.float32, device=dev)))
    sub = rng.choice([150, 600, 3000]); pb = rng.choice([1, 3, 12]); sg = rng.choice([1, 2, 4]); mf = rng.choice([500, 3000, 100000])
    try:
        with torch.no_grad():
            want = [patches.predict_mesh(net, p, f, sub_size=sub, n_iter=5, gt_points=g, patch_batch=pb) for p, f, g in lst]
            got = patches.predict_batch(net, lst, max_faces=mf, sub_size=sub, n_iter=5, patch_batch=pb, split_group=sg)
        for w, g in zip(want, got):
            same = all(torch.equal(torch.nan_to_num(w[k]), torch.nan_to_num(g[k])) for k in ('Vp', 'Np', 'V_updated'))
            same = same and w['n_patches'] == g['n_patches'] and (w['angle1'] == g['angle1'] or (w['angle1'] != w['angle1'] and g['angle1'] != g['angle1']))
            if not same:
                fails.append(('batch', [int(f.shape[0]) for _, f, _ in lst], sub, pb, sg, mf))
                break
    except Exception as ex:                                  # noqa: BLE001
        fails.append(('batch raised', [int(f.shape[0]) for _, f, _ in lst], sub, pb, sg, mf, repr(ex)[:300]))
    count['batch'] += 1
print('cases', count, 'failures', len(fails))
for f in fails[:12]:
    print('  FAIL', f)
sys.exit(1 if fails else 0)
