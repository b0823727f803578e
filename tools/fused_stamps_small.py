"""Phase timeline of the fused FeaSt kernel on a SMALL graph (the coarse levels' launches: fewer tiles than the chip has
slots) from a GEOBI_FUSED_STAMPS build:
   GEOBI_LIB=.../libgeobi_hip_stamps.so python tools/fused_stamps_small.py [freq cin cout fwd|dx]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import meshgen, _lib
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.feast_conv import FeaStConv
freq, cin, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (9, 128, 128)
what = sys.argv[4] if len(sys.argv) > 4 else 'fwd'
dev = torch.device('cuda:0')
pairs = [meshgen.synthetic_dual_data(freq, (0.1, 0.2, 0.3)[i % 3], seed=200 + i) for i in range(4)]
dv, df = union_batch(pairs)
df = df.to(dev)
g = df.graph(df.x.shape[0]).ensure_in()
N = df.x.shape[0]
conv = FeaStConv(cin, cout, 9).to(dev)
x = torch.randn(N, cin, device=dev, requires_grad=(what == 'dx'))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(4):
    if what == 'dx':
        y = conv(x, g, slope=0.2)
        y.backward(torch.ones_like(y))
    else:
        with torch.no_grad():
            ev[0].record(); conv(x, g, slope=0.2); ev[1].record()
torch.cuda.synchronize()
if what == 'fwd':
    print('%s %d->%d, N %d: events %.1f us' % (what, cin, cout, N, ev[0].elapsed_time(ev[1]) * 1e3))
if not hasattr(_lib.lib(), 'geobi_debug_stamps'):
    sys.exit(0)                      # the product library: timing only
buf = np.zeros((16384, 8), dtype=np.uint64)
rc = _lib.lib().geobi_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0
rows = 16 if os.environ.get('GEOBI_TILE16', '1') != '0' else 32
nb = (N + rows - 1) // rows
t = buf[:nb].astype(np.int64)
names = ['start->phase1 done', 'phase2 (row gathers)', 'store+issue B', 'barrier (tile published)', 'MFMA', 'barrier', 'reduce->epilogue']
d = np.diff(t, axis=1)
print('%s %d->%d, N %d, %d tiles of %d rows; 100 MHz ticks per phase (last chunk for chunked shapes), mean / median / p90' % (what, cin, cout, N, nb, rows))
for i, nme in enumerate(names):
    print('  %-28s %9.0f %9.0f %9.0f' % (nme, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
print('  whole tile %.0f; span first start -> last end %d ticks = %.1f us' % ((t[:, 7] - t[:, 0]).mean(), t[:, 7].max() - t[:, 0].min(), (t[:, 7].max() - t[:, 0].min()) / 100.0))
if what == 'fwd':
    print('  events: %.1f us' % (ev[0].elapsed_time(ev[1]) * 1e3))
