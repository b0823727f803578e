"""Inference driver counterpart of /root/reference/code/test_dual.py:90-150 (predict_dir) on synthetic
meshes: load a state dict written by tools/train_synthetic.py (reference key names), run every test mesh
through patches.predict_mesh (or, with --batched, patches.predict_batch; device preprocessing, patch split at --sub_size, network, merge, 60-sweep
vertex update) and report the per-mesh angular errors and their face-count-weighted means exactly as the
reference prints them.  The Synthetic test set is an external download, so the test list is SURVEY.md
section 8d's stand-in: 29 noisy icospheres, n drawn from {16, 22, 32, 45} x 3 noise levels.

  python tools/train_synthetic.py --max_epoch 12 --freq 16 --lr 2e-3 --out /tmp/net.pt
  python tools/test_synthetic.py --model /tmp/net.pt --sub_size 20000
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import network, meshgen, patches      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', type=str, default='', help='state dict (train_synthetic.py --out); random init if empty')
    ap.add_argument('--sub_size', type=int, default=20000)
    ap.add_argument('--n_meshes', type=int, default=29)
    ap.add_argument('--wei_param', type=int, default=2)
    ap.add_argument('--json', type=str, default='')
    ap.add_argument('--batched', action='store_true', help='the whole list through patches.predict_batch (small meshes as unions, '
                    'patch-split meshes in groups) instead of mesh by mesh: the same angles to the last bit, a list time instead of per-mesh times')
    opt = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = network.DualGNN(force_depth=False, pool_type='max', wei_param=opt.wei_param)
    if opt.model:
        net.load_state_dict(torch.load(opt.model, map_location='cpu', weights_only=True))
    net = net.to(dev).eval()
    freqs, sigmas = (16, 22, 32, 45), (0.1, 0.2, 0.3)
    err = np.zeros((3, opt.n_meshes))
    t_all = time.time()
    if opt.batched:
        data = [meshgen.noisy_icosphere(freqs[i % 4], sigmas[i % 3], seed=100 + i) for i in range(opt.n_meshes)]
        lst = [(torch.as_tensor(noisy, dtype=torch.float32, device=dev), torch.as_tensor(faces, dtype=torch.int32, device=dev),
                torch.as_tensor(clean, dtype=torch.float32, device=dev)) for noisy, clean, faces in data]
        torch.cuda.synchronize()
        t0 = time.time()
        res = patches.predict_batch(net, lst, sub_size=opt.sub_size, n_iter=60)
        torch.cuda.synchronize()
        print('predict_batch: %d meshes in %.4f s' % (opt.n_meshes, time.time() - t0), flush=True)
        for i, (r, (_, _, faces)) in enumerate(zip(res, data)):
            err[:, i] = faces.shape[0], r['angle1'], r['angle2']
            print("angle1: %9.6f,  angle2: %9.6f,  faces: %6d,  patches: %2d,  'ico%d_n%d'"
                  % (r['angle1'], r['angle2'], faces.shape[0], r['n_patches'], freqs[i % 4], 1 + i % 3), flush=True)
    for i in range(0 if opt.batched else opt.n_meshes):
        n, sg = freqs[i % 4], sigmas[i % 3]
        noisy, clean, faces = meshgen.noisy_icosphere(n, sg, seed=100 + i)
        t0 = time.time()
        r = patches.predict_mesh(net, noisy, faces, sub_size=opt.sub_size, n_iter=60, gt_points=clean)
        torch.cuda.synchronize()
        err[:, i] = faces.shape[0], r['angle1'], r['angle2']
        print("angle1: %9.6f,  angle2: %9.6f,  faces: %6d,  patches: %2d,  time: %7.4f s,  'ico%d_n%d'"
              % (r['angle1'], r['angle2'], faces.shape[0], r['n_patches'], time.time() - t0, n, 1 + i % 3), flush=True)
    count = err[0].sum()
    m1, m2 = (err[0] * err[1]).sum() / count, (err[0] * err[2]).sum() / count
    print("\nNum_face: %6d,  angle_mean1: %.6f,  angle_mean2: %.6f,  total %.2f s" % (count, m1, m2, time.time() - t_all))
    if opt.json:
        json.dump({'meshes': opt.n_meshes, 'faces': int(count), 'sub_size': opt.sub_size, 'angle_mean1_deg': m1,
                   'angle_mean2_deg': m2, 'model': os.path.basename(opt.model) or 'random init'}, open(opt.json, 'w'))


if __name__ == '__main__':
    main()
