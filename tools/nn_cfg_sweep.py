"""Tile-shape sweep of the NN GEMM on this path's shapes: run once per GEOBI_NN_CFG value (1: 64x64 k32,
2: 128x64, 4: 128x128, 5: 128x32, 6: 64x64 k64; unset: the library's own choice) and compare the lines.

  for c in "" 1 2 4 5 6; do GEOBI_NN_CFG=$c python tools/nn_cfg_sweep.py; done
"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import _lib as L
dev = torch.device('cuda:0'); lib = L.lib()
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
shapes = [(151380,32,576),(151380,64,312),(75692,64,312),(41000,128,600),(41000,64,1152),(41000,128,1152),(131072,128,1152),(131072,64,1152),(327680,32,576),(327680,64,312),(90000,128,600),(90000,64,1152)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    us = timeit(lambda: L.call('geobi_gemm_nn', L.ptr(A), K, L.ptr(B), N, 0, L.ptr(C), N, M, N, K, None, 1.0, L.stream()))
    print('cfg=%s M=%6d N=%5d K=%4d  %7.1f us  %5.1f TF' % (os.environ.get('GEOBI_NN_CFG', 'auto'), M, N, K, us, 2.0*M*N*K/us/1e6))
