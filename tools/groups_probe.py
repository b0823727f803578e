"""Concurrent mesh groups (executor.TrainGroups) against the single-union step: gradients, losses, ms per step.
usage: python tools/groups_probe.py [freq] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from geobi_gnn_amd import network, meshgen, executor
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.parallel import FlatParameters, batched_losses
from geobi_gnn_amd.train_util import FlatAdam

freq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net)
bucket = flat.bucket
opt = FlatAdam(flat.parameters(), lr=0.0)            # lr 0: every variant sees the same weights
pairs = [meshgen.synthetic_dual_data(freq, (0.1, 0.2, 0.3)[i % 3], seed=200 + i) for i in range(4)]


def union(idx):
    dv, df = union_batch([pairs[i] for i in idx]) if len(idx) > 1 else pairs[idx[0]]
    return dv.to(dev), df.to(dev)


dv, df = union([0, 1, 2, 3])


def ref_step():
    bucket.zero()
    a, b = dv.shallow_copy(), df.shallow_copy()
    vp, npred, _ = net((a, b))
    lv, ln = batched_losses(vp, npred, dv, df, 'L1', 'L1')
    (lv + ln).backward()
    opt.step()
    return lv.detach(), ln.detach()


for _ in range(5):
    lv, ln = ref_step()
torch.cuda.synchronize()
g_ref = bucket.flat.clone()
l_ref = (float(lv), float(ln))
t0 = time.perf_counter()
for _ in range(steps):
    ref_step()
torch.cuda.synchronize()
print('single union      : %.3f ms/step  loss %.6f %.6f' % ((time.perf_counter() - t0) / steps * 1e3, l_ref[0], l_ref[1]), flush=True)

for name, split in (('1 group of 4', [[0, 1, 2, 3]]), ('2 groups of 2', [[0, 1], [2, 3]]), ('2 groups swapped', [[2, 3], [0, 1]]),
                    ('4 groups of 1', [[0], [1], [2], [3]])):
    tg = executor.TrainGroups(net, bucket).set_groups([union(ix) for ix in split])

    def step():
        losses = tg.step()
        opt.step()
        return losses
    for _ in range(5):
        losses = step()
    torch.cuda.synchronize()
    g = bucket.flat.clone()
    ls = losses.sum(0).tolist()
    err = float((g - g_ref).abs().max() / g_ref.abs().max())
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print('%-18s: %.3f ms/step  loss %.6f %.6f  grad max err vs union %.2e (of max)  sequential fallbacks %d' %
          (name, ms, ls[0], ls[1], err, tg.sequential_steps), flush=True)
    if name == '2 groups of 2':
        g2 = g
    if name == '2 groups swapped':
        print('   A|B == B|A bitwise: %s' % bool(torch.equal(g, g2)), flush=True)
print('spin cap hits', executor.L.lib().geobi_net_spin_cap_hits())
