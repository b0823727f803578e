import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch, bench
from geobi_gnn_amd import network, executor
from geobi_gnn_amd.parallel import FlatParameters
from geobi_gnn_amd.train_util import FlatAdam
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev)
flat = FlatParameters(net); bucket = flat.bucket
opt = FlatAdam(flat.parameters(), lr=1e-3)
for rep in range(3):
    executor.STATS.update({'arena_retry': 0, 'fallback': 0, 'calls': 0})
    r = bench.measure_fresh_batch(net, bucket, opt, dev, 32)
    print(r['ms_per_step'], {k: executor.STATS.get(k) for k in ('arena_retry', 'fallback', 'calls', 'train_arena_bytes', 'train_need_bytes')}, flush=True)
