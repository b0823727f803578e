"""Robustness sweep: one training step (forward, losses, backward) on meshes of many sizes and on ragged
unions, to shake out size-dependent planning / workspace mistakes.  Prints one line per case."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import network, meshgen
from geobi_gnn_amd.data import union_batch
from geobi_gnn_amd.parallel import batched_losses

dev = torch.device('cuda:0')
torch.manual_seed(0)
net = network.DualGNN().to(dev)
cases = [[n] for n in (1, 2, 3, 5, 7, 10, 13, 20, 25, 40, 50, 64, 100, 128)]
cases += [[2, 40], [64, 3, 17], [1, 1, 1, 1], [100, 5], [32, 32, 32, 32, 32, 32, 32, 32]]
bad = 0
for ns in cases:
    try:
        duals = [meshgen.synthetic_dual_data(n, 0.2, seed=10 + i) for i, n in enumerate(ns)]
        dv, df = union_batch(duals) if len(duals) > 1 else duals[0]
        dv, df = dv.to(dev), df.to(dev)
        net.zero_grad(set_to_none=True)
        t0 = time.time()
        vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy()))
        if len(duals) > 1:
            lv, ln = batched_losses(vp, npred, dv, df, 'L1', 'L1')
        else:
            lv, ln = network.loss_v(vp, dv.y, 'L1'), network.loss_n(npred, df.y, 'L1')
        loss = network.dual_loss(lv, ln)
        loss.backward()
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(loss)) and all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
        print('n=%-28s faces=%8d  %7.1f ms  loss %.4f  %s' % (ns, df.x.shape[0], (time.time() - t0) * 1e3, float(loss),
                                                           'ok' if ok else 'NON-FINITE'), flush=True)
        bad += 0 if ok else 1
    except Exception as e:          # noqa: BLE001 -- report and keep sweeping
        bad += 1
        print('n=%-28s FAILED: %s' % (ns, str(e)[:200]), flush=True)
print('failures:', bad)
sys.exit(1 if bad else 0)
