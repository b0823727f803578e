import sys, os, torch
sys.path.insert(0, '/root/repo')
from geobi_gnn_amd import _lib as L
dev = torch.device('cuda:0'); lib = L.lib()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
def nn(M, N, K, transB=0):
    A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if transB else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    f = lambda: L.call('geobi_gemm_nn', L.ptr(A), K, L.ptr(B), B.shape[1], transB, L.ptr(C), N, M, N, K, None, 1.0, L.stream())
    us = timeit(f)
    print('NN  M=%7d N=%5d K=%5d blocks128=%6.2f/CU  %8.1f us  %6.1f TF/s  %6.0f GB/s' % (M, N, K, M/128/256, us, 2.0*M*N*K/us/1e6, 4.0*(M*K+K*N+M*N)/us*1e-3))
for M in (32768, 65536, 81920, 98304, 131072, 262144):
    nn(M, 32, 576)
for M in (32768, 65536, 131072):
    nn(M, 64, 1152)
    nn(M, 128, 1152)
