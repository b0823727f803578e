"""Micro-benchmark of the dense kernels on the shapes the bench workload launches (run on the GPU box)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import _lib as L

dev = torch.device('cuda:0')
lib = L.lib()

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us

def nn(M, N, K, transB=0):
    A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if transB else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    f = lambda: L.call('geobi_gemm_nn', L.ptr(A), K, L.ptr(B), B.shape[1], transB, L.ptr(C), N, M, N, K, None, 1.0, L.stream())
    us = timeit(f)
    gb = 4.0 * (M * K + K * N + M * N) / 1e9
    print('NN  M=%6d N=%5d K=%5d tB=%d  %8.1f us  %6.1f TF/s  %6.0f GB/s' % (M, N, K, transB, us, 2.0 * M * N * K / us / 1e6, gb / us * 1e6))

def tn(M, I, J):
    A = torch.randn(M, I, device=dev); B = torch.randn(M, J, device=dev); C = torch.empty(I, J, device=dev)
    ws = L.workspace(lib.geobi_gemm_tn_ws_bytes(I, J, M), dev)
    f = lambda: L.call('geobi_gemm_tn', L.ptr(A), I, L.ptr(B), J, M, I, J, L.ptr(C), J, L.ptr(ws), ws.numel(), L.stream())
    us = timeit(f)
    gb = 4.0 * (M * I + M * J) / 1e9
    print('TN  M=%6d I=%5d J=%5d       %8.1f us  %6.1f TF/s  %6.0f GB/s' % (M, I, J, us, 2.0 * M * I * J / us / 1e6, gb / us * 1e6))

N0, N1, N2 = 81920, 22000, 6000
ONLY_TN = len(sys.argv) > 1 and sys.argv[1] == 'tn'
_nn = nn
if ONLY_TN:
    nn = lambda *a, **k: None
print('--- forward out = z Wf')
for M, N, K in [(N0, 32, 108), (N0, 32, 576), (N1, 64, 288), (N1, 64, 1152), (N2, 128, 576), (N2, 128, 1152)]: nn(M, N, K)
print('--- backward dz = g Wf^T')
for M, N, K in [(N0, 576, 32), (N0, 108, 32), (N1, 1152, 64), (N1, 288, 64), (N2, 576, 128), (N2, 1152, 128)]: nn(M, N, K, 1)
print('--- backward dx = r W')
for M, N, K in [(N0, 64, 312), (N0, 12, 312), (N1, 128, 600), (N1, 32, 600), (N2, 64, 1176), (N2, 128, 1176)]: nn(M, N, K)
print('--- heads')
nn(N0, 1024, 32, 1); nn(N0, 32, 1024)
print('--- weight gradients')
for M, I, J in [(N0, 576, 32), (N0, 108, 32), (N1, 1152, 64), (N1, 288, 64), (N2, 576, 128), (N2, 1152, 128),   # z^T g (the model adds an implicit ones row)
                (N0, 24, 65), (N0, 24, 33), (N0, 24, 13), (N1, 24, 129), (N1, 24, 33), (N2, 24, 65), (N2, 24, 129),
                (N0, 3, 1025), (N0, 1024, 33)]: tn(M, I, J)
