"""Phase cycles of the channel-chunked backward kernel (feast_rowpass_fused128_kernel) from a GEOBI_FUSED_STAMPS build:
   GEOBI_LIB=.../libgeobi_hip_stamps.so python tools/rp128_stamps.py [freq cin cout]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import meshgen, _lib
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.feast_conv import FeaStConv
freq, cin, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 128, 64)
dev = torch.device('cuda:0')
pairs = [meshgen.synthetic_dual_data(freq, (0.1, 0.2, 0.3)[i % 3], seed=200 + i) for i in range(4)]
dv, df = union_batch(pairs)
df = df.to(dev)
g = df.graph(df.x.shape[0]).ensure_in()
N = df.x.shape[0]
conv = FeaStConv(cin, cout, 9).to(dev)
x = torch.randn(N, cin, device=dev, requires_grad=True)
for _ in range(4):
    y = conv(x, g, slope=0.2)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
buf = np.zeros((16384, 8), dtype=np.uint64)
rc = _lib.lib().geobi_debug_stamps_bwd(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0
nb = (N + 31) // 32
t = buf[:nb, :4].astype(np.int64)
names = ['g tile + setup', 'matrix phases (all chunks)', 'row passes (all chunks)', 'softmax backward']
print('%d->%d backward, N %d, %d tiles of 32 rows; shader cycles of thread 0 per phase incl. the barriers, mean / median / p90' % (cin, cout, N, nb))
for i, nme in enumerate(names):
    print('  %-28s %9.0f %9.0f %9.0f' % (nme, t[:, i].mean(), np.median(t[:, i]), np.percentile(t[:, i], 90)))
print('  whole tile %.0f' % t.sum(1).mean())
