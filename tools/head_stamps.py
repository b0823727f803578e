"""Where wave 0 of each block of the backward head kernel spends its time (GEOBI_HEAD_STAMPS build, diagnostic):
   GEOBI_LIB=.../libgeobi_hip_hstamps.so python tools/head_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import _lib as L
dev = torch.device('cuda:0')
N = 81920
torch.manual_seed(0)
x = torch.randn(N, 32, device=dev)
w1 = torch.randn(1024, 32, device=dev) * 0.1; b1 = torch.randn(1024, device=dev) * 0.1
w2 = torch.randn(3, 1024, device=dev) * 0.1
raw = torch.randn(N, 3, device=dev); graw = torch.randn(N, 3, device=dev)
dx = torch.empty(N, 32, device=dev)
dw1 = torch.zeros_like(w1); db1 = torch.zeros_like(b1); dw2 = torch.zeros_like(w2); db2 = torch.zeros(3, device=dev)
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
for _ in range(3):
    L.call('geobi_head_bwd', L.ptr(x), 32, N, L.ptr(w1), L.ptr(b1), 1024, L.ptr(w2), 3, 0.2, 1, None, None, L.ptr(raw),
           L.ptr(graw), L.ptr(dx), L.ptr(dw1), L.ptr(db1), L.ptr(dw2), L.ptr(db2), 0, L.ptr(ws), ws.numel(), L.stream())
torch.cuda.synchronize()
buf = np.zeros((1024, 8), dtype=np.uint64)
assert L.lib().geobi_debug_head_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
t = buf[:256].astype(np.float64)
names = ['tile prologue', 'issue recompute chain', 'request operands, wait h, form dh', 'issue dW1 chain, transpose dh',
         'issue dx chain', 'db1 / dW2 sums, dW1 tile added to LDS', 'dx fold + stores', '-']
tot = t.sum(axis=1).mean()
print('wave 0 of a block: %.0f ticks in all (10 tiles x 8 chunks); share per phase' % tot)
for i, n in enumerate(names[:7]):
    print('  %-40s %9.0f  %5.1f %%' % (n, t[:, i].mean(), 100 * t[:, i].mean() / tot))
