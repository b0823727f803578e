"""Randomised parity sweep, second part (the first is tools/fuzz_kernels.py): the pieces around the FeaSt layers and the whole
network on random sizes -- pooled-edge construction and segment max / mean / unpool against the oracle; the fused heads at
random row counts (ragged last tiles); the whole DualGNN (own matching, the oracle replays its clusters) forward + backward on
noisy icospheres of random frequency, noise and weights, single meshes and unions; the patch split against the sequential
statement on random meshes and patch sizes.     python tools/fuzz_model.py [seconds] [seed]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_gpu_kernels as K
import test_gpu_model as M
import test_gpu_patches as TP
from helpers import rel_err
from geobi_gnn_amd import net_util, ops, network, meshgen, patches
from geobi_gnn_amd.data import union_batch
from oracle import ref_model as R, pyg_ops as P, mesh_ops
from oracle.weights import make_state_dict

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda:0')
torch.set_num_threads(8)
count = {'pool_edge': 0, 'segments': 0, 'heads': 0, 'model': 0, 'split': 0}
fails = []
ucases = [0]
t_end = time.time() + budget


def check(tag, ok, detail):
    if not ok:
        fails.append((tag, detail))


while time.time() < t_end:
    seed = rng.randrange(1 << 30)
    g = torch.Generator().manual_seed(seed)
    # ---- pool_edge on a random symmetric graph and a random consecutive clustering
    n = rng.choice([2, 10, 300, 2000, 9000])
    ei = K._sym_graph(n, rng.choice([n, 4 * n]), seed=seed % 100000)
    if ei.shape[1]:
        w = torch.rand(ei.shape[1], generator=g)
        cluster = torch.unique(torch.randint(0, max(n // rng.choice([1, 2, 3]), 1), (n,), generator=g), return_inverse=True)[1]
        ref_i, ref_w = R.pool_edge(cluster, ei, w)
        out_i, out_w = net_util.pool_edge(cluster.to(dev), ei.to(dev), w.to(dev))
        check('pool_edge', torch.equal(out_i.cpu(), ref_i) and (ref_w.numel() == 0 or rel_err(out_w.cpu(), ref_w.double()) < 1e-6), (n, seed))
        count['pool_edge'] += 1
    # ---- segment max / mean and the unpool gather
    n = rng.choice([1, 17, 1000, 5000]); nseg = max(1, n // rng.choice([1, 2, 4])); C = rng.choice([32, 64, 128])
    seg = torch.randint(0, nseg, (n,), generator=g)
    x = torch.randn(n, C, generator=g); gout = torch.randn(nseg, C, generator=g)
    sidx = ops.SegmentIndex(seg.to(torch.int32).to(dev), nseg)
    for red, fn in (('max', ops.SegmentMaxFn), ('mean', ops.SegmentMeanFn)):
        xo = x.clone().requires_grad_(True)
        ref = P.scatter(xo, seg, dim=0, dim_size=nseg, reduce=red); ref.backward(gout)
        xh = x.to(dev).requires_grad_(True)
        out = fn.apply(xh, sidx); out.backward(gout.to(dev))
        if red == 'max':
            check('segment max', torch.equal(out.detach().cpu(), ref.detach()) and torch.equal(xh.grad.cpu(), xo.grad), (n, nseg, C, seed))
        else:
            check('segment mean', rel_err(out.detach().cpu(), ref.detach()) < K.TOL and rel_err(xh.grad.cpu(), xo.grad) < K.TOL, (n, nseg, C, seed))
    count['segments'] += 1
    # ---- fused heads at a random row count
    N = rng.choice([1, 31, 32, 33, 500, 4097]); mode = rng.choice([0, 1]); nout = 3 if mode == 1 else rng.choice([1, 3])
    torch.manual_seed(seed % 1000)
    fc1, fc2 = torch.nn.Linear(32, 1024).double(), torch.nn.Linear(1024, nout).double()
    xx = torch.randn(N, 32, dtype=torch.double); x6 = torch.randn(N, 6, dtype=torch.double)
    dd = torch.nn.functional.normalize(torch.randn(N, 3, dtype=torch.double), dim=1); go = torch.randn(N, 3, dtype=torch.double)
    xo = xx.clone().requires_grad_(True)
    y = fc2(torch.nn.functional.leaky_relu(fc1(xo), 0.2))
    ref = (y * dd if nout == 1 else y) + x6[:, :3] if mode == 0 else torch.nn.functional.normalize(y, dim=1)
    ref.backward(go)
    f = lambda t: t.detach().float().to(dev)
    xh = f(xx).requires_grad_(True)
    ps = [f(p).requires_grad_(True) for p in (fc1.weight, fc1.bias, fc2.weight, fc2.bias)]
    out = ops.HeadFn.apply(xh, ps[0], ps[1], ps[2], ps[3], mode, f(dd) if (mode == 0 and nout == 1) else None, f(x6) if mode == 0 else None)
    out.backward(f(go))
    e = max([rel_err(out.detach().cpu(), ref.detach()), rel_err(xh.grad.cpu(), xo.grad)] +
            [rel_err(ph.grad.cpu(), po.grad) for ph, po in zip(ps, (fc1.weight, fc1.bias, fc2.weight, fc2.bias))])
    if not e < K.TOL:
        # the same head in fp32 on the CPU: is the distance to fp64 the arithmetic's (a hidden pre-activation within rounding of
        # zero takes the other branch of the leaky-relu) or the kernel's?
        f1, f2 = torch.nn.Linear(32, 1024), torch.nn.Linear(1024, nout)
        f1.load_state_dict({k: v.float() for k, v in fc1.state_dict().items()}); f2.load_state_dict({k: v.float() for k, v in fc2.state_dict().items()})
        xc = xx.float().requires_grad_(True)
        yc = f2(torch.nn.functional.leaky_relu(f1(xc), 0.2))
        rc = (yc * dd.float() if nout == 1 else yc) + x6[:, :3].float() if mode == 0 else torch.nn.functional.normalize(yc, dim=1)
        rc.backward(go.float())
        e_cpu = max([rel_err(rc.detach(), ref.detach()), rel_err(xc.grad, xo.grad)] +
                    [rel_err(pc.grad, po.grad) for pc, po in zip((f1.weight, f1.bias, f2.weight, f2.bias), (fc1.weight, fc1.bias, fc2.weight, fc2.bias))])
        near = int((fc1(xx).abs() < 1e-6).sum())                # hidden pre-activations within fp32 rounding of the kink
        if e_cpu > 0.2 * e or near > 0:           # within 5 x the CPU's own fp32 distance to fp64, or a kink in reach
            count['heads: fp32 arithmetic'] = count.get('heads: fp32 arithmetic', 0) + 1
        else:
            check('heads', False, (N, mode, nout, seed, e, 'cpu fp32: %.1e, pre-activations near zero %d' % (e_cpu, near)))
    count['heads'] += 1
    # ---- the whole network: own matching on the device, the oracle replays the clusters
    freqs = [rng.choice([2, 3, 4, 5, 6, 7, 9])] if rng.random() < 0.6 else [rng.choice([2, 3, 4, 5]) for _ in range(rng.choice([2, 3]))]
    sd = make_state_dict(R.DualGNN().state_dict(), seed % 97)
    net = M._hip_net(sd, dev)
    pairs = []
    for i, fq in enumerate(freqs):
        if fq >= 4 and rng.random() < 0.4:                   # an irregular mesh: holes, boundaries, vertices without faces
            noisy, clean, faces = meshgen.noisy_icosphere(fq, rng.choice([0.1, 0.3]), (seed + i) % 100000)
            faces = np.ascontiguousarray(faces[np.random.default_rng(seed + i).random(faces.shape[0]) >= rng.choice([0.05, 0.2])])
            pairs.append(meshgen.build_dual_data(noisy, faces, clean, name='m'))
        else:
            pairs.append(meshgen.synthetic_dual_data(fq, rng.choice([0.1, 0.2, 0.3]), seed=(seed + i) % 100000))
    dv, df = pairs[0] if len(pairs) == 1 else union_batch(pairs)
    dvo = P.Data(dv.x.clone(), dv.edge_index.clone(), edge_weight=dv.edge_weight.clone(), y=dv.y.clone())
    dfo = P.Data(df.x.clone(), df.edge_index.clone(), edge_weight=df.edge_weight.clone(), y=df.y.clone(), fv_indices=df.fv_indices.clone())
    try:
        vp, npred, loss, err_n = M._step(net, network, dv.to(dev), df.to(dev))
        raw = []
        for m in (net.gnn_v.pooling1, net.gnn_v.pooling2, net.gnn_f.pooling1, net.gnn_f.pooling2):
            raw += [c.cpu() for c in m.last_clusters]
        ora = R.DualGNN(); ora.load_state_dict(sd)
        M.install_replay(ora, raw)
        vo, no, loss_o, err_o = M._step(ora, R, dvo, dfo)
        ok = rel_err(vp.cpu(), vo) < M.OUT_TOL and rel_err(npred.cpu(), no) < M.OUT_TOL and abs(loss - loss_o) < 1e-5 * abs(loss_o)
        worst = ''
        for (k, ph), (_, po) in zip(net.named_parameters(), ora.named_parameters()):
            if not rel_err(ph.grad.cpu(), po.grad) < M._grad_tol(k):
                ok, worst = False, k
        if not ok:
            # `ora` above is the oracle in fp32 on the CPU (the tests' comparison).  Against the oracle in fp64 with the same
            # clusters: is the device further from the truth than the CPU's own fp32 arithmetic is?
            o64 = R.DualGNN().double(); o64.load_state_dict({k: v.double() for k, v in sd.items()})
            M.install_replay(o64, raw)
            near_zero = [0]

            def hook(mod, inp, outp):
                near_zero[0] += int((outp.detach().abs() < 1e-6).sum())
            hs = [m.register_forward_hook(hook) for nme, m in o64.named_modules()
                  if isinstance(m, P.FeaStConv) or nme in ('fc_v1', 'fc_f1')]
            dd_v = P.Data(dv.x.double(), dv.edge_index.clone(), edge_weight=dv.edge_weight.double(), y=dv.y.double())
            dd_f = P.Data(df.x.double(), df.edge_index.clone(), edge_weight=df.edge_weight.double(), y=df.y.double(), fv_indices=df.fv_indices.clone())
            v64, n64, _, _ = M._step(o64, R, dd_v, dd_f)
            for h_ in hs:
                h_.remove()
            worse = []
            for (k, ph), (_, po), (_, p64) in zip(net.named_parameters(), ora.named_parameters(), o64.named_parameters()):
                eh, ec = rel_err(ph.grad.cpu(), p64.grad), rel_err(po.grad, p64.grad)
                if eh > max(M._grad_tol(k), 3.0 * ec):
                    if k.endswith('u.weight') and ec > 1e-5:
                        ucases[0] += 1           # a sum of cancelling terms the CPU's fp32 cannot hold to the bar either
                        continue
                    worse.append((k, '%.1e' % eh, 'cpu fp32 %.1e' % ec))
            eo = max(rel_err(vp.cpu(), v64), rel_err(npred.cpu(), n64))
            eo_c = max(rel_err(vo, v64), rel_err(no, n64))
            if worse and near_zero[0] > 0 and eo < max(M.OUT_TOL, 3.0 * eo_c):
                # a pre-activation within rounding of a leaky-relu's kink: which branch fp32 takes is the summation order's
                count['model: a pre-activation at a kink'] = count.get('model: a pre-activation at a kink', 0) + 1
            elif not worse and eo < max(M.OUT_TOL, 3.0 * eo_c):
                count['model: within fp32 arithmetic'] = count.get('model: within fp32 arithmetic', 0) + 1
            else:
                check('model', False, (freqs, seed, 'outputs %.1e (cpu fp32 %.1e)' % (eo, eo_c), worse[:4]))
    except Exception as ex:                                 # noqa: BLE001 -- a sweep reports and goes on
        fails.append(('model raised', (freqs, seed, repr(ex)[:300])))
    count['model'] += 1
    # ---- the patch split against the sequential statement
    fq = rng.choice([4, 6, 9, 12]); sub = rng.choice([50, 333, 1000, 5000])
    noisy, _, faces = meshgen.noisy_icosphere(fq, 0.2, seed=seed % 100000)
    pts = torch.as_tensor(noisy, dtype=torch.float32, device=dev); fv = torch.as_tensor(faces, dtype=torch.int32, device=dev)
    got = [s.cpu().numpy() for s in patches.split_faces(pts, fv, sub)]
    rp, ls = TP._host_incidence(faces, pts.shape[0])
    d2 = ((pts[fv.long()].mean(1) - pts.mean(0, keepdim=True)) ** 2).sum(1).cpu().numpy()
    want = mesh_ops.split_faces(d2, np.ascontiguousarray(faces, dtype=np.int32), rp, ls, sub)
    check('split', len(got) == len(want) and all(np.array_equal(a, b) for a, (_, b) in zip(got, want)), (fq, sub, seed))
    count['split'] += 1
print('cases', count, '| deep u.weight gradients beyond 3 x the CPU fp32 error where that is itself > 1e-5:', ucases[0], '| failures', len(fails))
for f in fails[:12]:
    print('  FAIL', f)
sys.exit(1 if fails else 0)
