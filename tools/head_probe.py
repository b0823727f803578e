"""Time the fused regression heads alone (forward, backward) at the bench's facet count:
   GEOBI_LIB=<variant build> python tools/head_probe.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import _lib as L

dev = torch.device('cuda:0')
N = 81920
torch.manual_seed(0)
x = torch.randn(N, 32, device=dev)
w1 = torch.randn(1024, 32, device=dev) * 0.1; b1 = torch.randn(1024, device=dev) * 0.1
w2 = torch.randn(3, 1024, device=dev) * 0.1; b2 = torch.randn(3, device=dev) * 0.1
raw = torch.empty(N, 3, device=dev); out = torch.empty(N, 3, device=dev)
graw = torch.randn(N, 3, device=dev)
dx = torch.empty(N, 32, device=dev)
dw1 = torch.zeros_like(w1); db1 = torch.zeros_like(b1); dw2 = torch.zeros_like(w2); db2 = torch.zeros_like(b2)
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
st = L.stream()
def fwd():
    L.call('geobi_head_fwd', L.ptr(x), 32, N, L.ptr(w1), L.ptr(b1), 1024, L.ptr(w2), L.ptr(b2), 3, 0.2, 1, None, None, 0,
           None, L.ptr(raw), L.ptr(out), st)
def bwd():
    L.call('geobi_head_bwd', L.ptr(x), 32, N, L.ptr(w1), L.ptr(b1), 1024, L.ptr(w2), 3, 0.2, 1, None, None, L.ptr(raw),
           L.ptr(graw), L.ptr(dx), L.ptr(dw1), L.ptr(db1), L.ptr(dw2), L.ptr(db2), 0, L.ptr(ws), ws.numel(), st)
res = {}
for name, f in (('fwd', fwd), ('bwd', bwd)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    res[name + ' us'] = round(a.elapsed_time(b) * 50, 1)
print(json.dumps(res))
