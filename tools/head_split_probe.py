"""The fused heads in exact fp32 against the split-precision form (geobi_set_head_precision): time per launch at the bench's
facet count and distance of both to an fp64 evaluation.   python tools/head_split_probe.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import _lib as L

dev = torch.device('cuda:0')
N = 81920
torch.manual_seed(0)
x = torch.randn(N, 32, device=dev)
w1 = (torch.rand(1024, 32, device=dev) * 2 - 1) / 32 ** 0.5; b1 = (torch.rand(1024, device=dev) * 2 - 1) / 32 ** 0.5
w2 = (torch.rand(3, 1024, device=dev) * 2 - 1) / 32; b2 = (torch.rand(3, device=dev) * 2 - 1) / 32
graw = torch.randn(N, 3, device=dev) * 1e-3
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
st = L.stream()
lib = L.lib()

def run(mode_id):
    raw = torch.empty(N, 3, device=dev); out = torch.empty(N, 3, device=dev); dx = torch.empty(N, 32, device=dev)
    dw1 = torch.zeros_like(w1); db1 = torch.zeros_like(b1); dw2 = torch.zeros_like(w2); db2 = torch.zeros_like(b2)
    def fwd():
        L.call('geobi_head_fwd', L.ptr(x), 32, N, L.ptr(w1), L.ptr(b1), 1024, L.ptr(w2), L.ptr(b2), 3, 0.2, 1, None, None, 0,
               None, L.ptr(raw), L.ptr(out), st)
    def bwd():
        L.call('geobi_head_bwd', L.ptr(x), 32, N, L.ptr(w1), L.ptr(b1), 1024, L.ptr(w2), 3, 0.2, 1, None, None, L.ptr(raw),
               L.ptr(graw), L.ptr(dx), L.ptr(dw1), L.ptr(db1), L.ptr(dw2), L.ptr(db2), 0, L.ptr(ws), ws.numel(), st)
    lib.geobi_set_head_precision(mode_id)
    t = {}
    for name, f in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        t[name + '_us'] = round(a.elapsed_time(b) * 50, 1)
    lib.geobi_set_head_precision(0)
    return t, dict(raw=raw, out=out, dx=dx, dw1=dw1, db1=db1, dw2=dw2, db2=db2)

# fp64 reference (mode 1 head: out = normalize(raw)); the backward takes the gradient w.r.t. `raw`
xd, w1d, b1d, w2d, b2d = (t.double().cpu().requires_grad_(True) for t in (x, w1, b1, w2, b2))
pre = xd @ w1d.t() + b1d
h = torch.nn.functional.leaky_relu(pre, 0.2)
rawd = h @ w2d.t() + b2d
rawd.backward(graw.double().cpu())
want = dict(raw=rawd.detach(), out=torch.nn.functional.normalize(rawd.detach(), dim=1), dx=xd.grad, dw1=w1d.grad, db1=b1d.grad,
            dw2=w2d.grad, db2=b2d.grad)
res = {}
for mode_id, name in ((0, 'fp32'), (1, 'bf16x3 (six products)')):
    t, got = run(mode_id)
    err = {k: float((got[k].double().cpu() - want[k]).abs().max() / want[k].abs().max()) for k in want}
    res[name] = {'time': t, 'max_err_over_max_ref': {k: '%.2e' % v for k, v in err.items()}}
print(json.dumps(res, indent=1))
