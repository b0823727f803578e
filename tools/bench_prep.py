"""Mesh -> network inputs: device path (meshprep) beside the host generator (meshgen, numpy) on the
BASELINE.json mesh sizes.  Prints one JSON line per size.

  python tools/bench_prep.py            # n = 32 (20k faces) and n = 87 (151k faces)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import meshgen, meshprep      # noqa: E402


def main():
    dev = torch.device('cuda:0')
    for n in (32, 87):
        noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=n)
        t0 = time.time()
        meshgen.build_dual_data(noisy, faces, clean)
        host_s = time.time() - t0
        pts = torch.from_numpy(noisy).to(dev)
        gt = torch.from_numpy(clean).to(dev)
        fv = torch.from_numpy(faces).to(dev).int()
        for _ in range(2):
            meshprep.build_dual_data(pts, fv, gt, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 10
        for _ in range(reps):
            dv, df = meshprep.build_dual_data(pts, fv, gt, device=dev)
        torch.cuda.synchronize()
        dev_s = (time.time() - t0) / reps
        print(json.dumps({'n': n, 'faces': int(faces.shape[0]), 'vertices': int(noisy.shape[0]),
                          'edges_v': int(dv.graph().E), 'edges_f': int(df.graph().E),
                          'host_numpy_ms': round(host_s * 1e3, 1), 'device_ms': round(dev_s * 1e3, 3),
                          'note': 'device time includes 5 host read-backs (2 edge counts, max valence, scale, index check)'}),
              flush=True)


if __name__ == '__main__':
    main()
