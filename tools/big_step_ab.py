import json, os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from geobi_gnn_amd import network, meshgen, meshprep, executor
dev = torch.device('cuda:0')
n = int(sys.argv[1])
noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=1)
pts, gt = torch.from_numpy(noisy).to(dev), torch.from_numpy(clean).to(dev)
fv = torch.from_numpy(faces).to(dev).int()
torch.manual_seed(0)
net = network.DualGNN().to(dev)
dv, df = meshprep.build_dual_data(pts, fv, gt, device=dev)
for en in (True, False, True, False):
    executor.ENABLED = en
    for rep in range(3):
        net.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); t0 = time.time()
        vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy()))
        torch.cuda.synchronize(); t1 = time.time()
        loss = network.dual_loss(network.loss_v(vp, dv.y, 'L1'), network.loss_n(npred, df.y, 'L1'))
        loss.backward()
        torch.cuda.synchronize(); t2 = time.time()
        print('executor=%s rep %d: fwd %.1f ms, bwd %.1f ms' % (en, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
print(executor.STATS)
