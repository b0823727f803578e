import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import _lib as L
dev = torch.device('cuda:0')
rowptr = torch.zeros(2, dtype=torch.int32, device=dev); row = torch.zeros(4, dtype=torch.int32, device=dev)
lib = L.lib()
f = lib.geobi_expand_rowptr
def run(n, stream):
    rp, r = rowptr.data_ptr(), row.data_ptr()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f(rp, 1, r, stream)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print('null stream      : host %.2f us/launch, incl. drain %.2f' % run(20000, 0))
s = torch.cuda.Stream()
print('torch side stream: host %.2f us/launch, incl. drain %.2f' % run(20000, s.cuda_stream))
x = torch.empty(1024, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20000): x.add_(1.0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('torch add_ (null): host %.2f us/op' % ((t1 - t0) / 20000 * 1e6))
with torch.cuda.stream(s):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20000): x.add_(1.0)
    t1 = time.perf_counter(); torch.cuda.synchronize()
print('torch add_ (side): host %.2f us/op' % ((t1 - t0) / 20000 * 1e6))
t0 = time.perf_counter()
for _ in range(20000): y = torch.empty(4096, device=dev)
print('torch.empty cuda : %.2f us' % ((time.perf_counter() - t0) / 20000 * 1e6))
c = torch.zeros(4, dtype=torch.int32, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): c.tolist()
print('tolist (idle GPU): %.2f us' % ((time.perf_counter() - t0) / 2000 * 1e6))
