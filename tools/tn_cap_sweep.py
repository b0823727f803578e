"""Slab-count sweep of the TN (weight-gradient) GEMM: run once per GEOBI_TN_CAP value (blocks per launch the
planner aims for; unset: the library's own choice) and compare.  Times include the slab reduction.

  for c in 64 128 256 512 1024; do GEOBI_TN_CAP=$c python tools/tn_cap_sweep.py; done
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
for (M, I, J) in [(81920,24,65),(81920,24,33),(81920,24,13),(40968,24,65),(22000,24,129),(22000,24,65),(11000,24,129),(6000,24,129),(6000,24,65),(81920,576,32),(40968,576,32),(22000,1152,64),(6000,1152,128),(81920,108,32)]:
    A = torch.randn(M, I, device=dev); B = torch.randn(M, J, device=dev); C = torch.empty(I, J, device=dev)
    ws = L.workspace(lib.geobi_gemm_tn_ws_bytes(I, J, M), dev)
    us = timeit(lambda: L.call('geobi_gemm_tn', L.ptr(A), I, L.ptr(B), J, M, I, J, L.ptr(C), J, L.ptr(ws), ws.numel(), L.stream()))
    print('cap=%s M=%6d I=%5d J=%4d  %7.1f us' % (os.environ.get('GEOBI_TN_CAP', '512'), M, I, J, us))
