"""Phase timeline of the fused backward row-pass kernel from a GEOBI_FUSED_STAMPS build (diagnostic):
   GEOBI_LIB=.../libgeobi_hip_stamps.so python tools/k2_stamps.py [cin cout]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import meshgen, _lib
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.feast_conv import FeaStConv
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 32)
dev = torch.device('cuda:0')
pairs = [meshgen.synthetic_dual_data(32, (0.1, 0.2, 0.3)[i % 3], seed=200 + i) for i in range(4)]
dv, df = union_batch(pairs)
df = df.to(dev)
g = df.graph(df.x.shape[0]).ensure_in()
N = df.x.shape[0]
conv = FeaStConv(cin, cout, 9).to(dev)
x = torch.randn(N, cin, device=dev, requires_grad=True)
gout = torch.randn(N, cout, device=dev)
for _ in range(3):
    conv(x, g, slope=0.2).backward(gout)
torch.cuda.synchronize()
buf = np.zeros((16384, 8), dtype=np.uint64)
lib = _lib.lib()
rc = lib.geobi_debug_stamps_bwd(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0
rows = 16 if os.environ.get('GEOBI_TILE16', '1') != '0' else 32      # tile geometry of the build / switch
nb = (N + rows - 1) // rows
t = buf[:nb, :5].astype(np.int64)
names = ['g tile + barrier', 'MFMA (wave 0)', 'barrier', 'row pass incl. sums (lane 0)']
d = np.diff(t, axis=1)
print('%d-row tiles; ' % rows, end='')
print('layer %d->%d backward, %d tiles; s_memtime ticks per phase, mean / median / p90' % (cin, cout, nb))
for i, nme in enumerate(names):
    print('  %-28s %9.0f %9.0f %9.0f' % (nme, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
print('  %-28s %9.0f' % ('whole tile (wave 0)', (t[:, 4] - t[:, 0]).mean()))
f = buf[:nb, :8].astype(np.int64)
if (f[:, 5:8] > 0).all():        # staged row pass (64 channels): its own steps, first chunk of items of wave 0's nodes
    seq = np.stack([f[:, 3], f[:, 5], f[:, 6], f[:, 7], f[:, 4]], 1)
    dd = np.diff(seq, axis=1)
    for i, nme in enumerate(['  dz pieces, ids, logits, softmax', '  wait: first half rows', '  FMAs + wait: second half rows',
                             '  FMAs, dl rows, node sums']):
        print('  %-34s %9.0f %9.0f %9.0f' % (nme, dd[:, i].mean(), np.median(dd[:, i]), np.percentile(dd[:, i], 90)))
span = t[:, 4].max() - t[:, 0].min()
print('  span first start -> last end: %d ticks; tiles in flight on average: %.1f' % (span, (t[:, 4] - t[:, 0]).sum() / span))
