"""Accuracy of the softmax's exp on the device against fp64: single FeaSt layers (64 -> 32, 12 -> 32 with |x| ~ 30) and the
whole network at n = 32; prints the maximal relative errors of outputs and gradients (run under two builds to compare:
GEOBI_LIB=<variant with -DGEOBI_EXP_OCML>)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_kernels as K
dev = torch.device('cuda:0')
res = {}
for (cin, cout, xs) in ((64, 32, 1.0), (32, 64, 1.0), (128, 64, 1.0), (12, 32, 30.0), (64, 32, 8.0)):
    ei = K._sym_graph(3000, 12000, seed=cin + cout)
    errs = K._run_feast(dev, cin, cout, ei, 3000, 0.2, False, seed=5, xscale=xs)
    res['%d->%d x%g' % (cin, cout, xs)] = {k: float('%.3g' % v) for k, v in errs.items()}
print(json.dumps(res, indent=1))
