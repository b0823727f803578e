"""decide the split-precision question with numbers before building it.

The fp32 MFMA rate of gfx950 is 1/16 of the bf16 rate.  A product of two fp32 matrices can be formed from bf16 pieces
with fp32 accumulation:   a = a1 + a2 (+ a3),  b = b1 + b2 (+ b3),  every piece a bf16 number (round-to-nearest of what
is left), so that every partial product is exact in the fp32 accumulator.
    2-way / 3 products:  a1 b1 + a1 b2 + a2 b1                                   (~16 mantissa bits)
    3-way / 6 products:  a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2           (~24 mantissa bits)
This script measures, on the REAL operands of the path, the distance to an fp64 product of: plain fp32, 1 bf16 product,
the 3- and the 6-product forms -- for the two head GEMMs (network.py:309-316,324-341: 32 -> 1024 -> 3) and the node
transform of one 128 -> 64 layer (gnn_f.r_conv1, level 1).  Operands come from a forward pass of the product on the
bench's mesh size (n = 32, random-init weights = the bench's; --trained PATH: a state dict trained by
tools/train_synthetic.py), captured with module hooks; the emulation itself is CPU torch (bf16-valued fp32 matrices,
fp32 matmul: products exact, accumulation fp32 like the MFMA's).

    python tools/bf16_split_study.py [--freq 32] [--trained net.pt] > profiles/r04_bf16_split_study.txt
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def pieces(a, k):
    """a (fp32) -> k bf16-valued fp32 tensors with a ~= sum(pieces)."""
    out, rest = [], a.clone()
    for _ in range(k):
        p = rest.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        rest = rest - p
    return out


def split_matmul(a, b, ways, terms):
    pa, pb = pieces(a, ways), pieces(b, ways)
    acc = torch.zeros(a.shape[0], b.shape[1], dtype=torch.float32)
    for i, j in terms:                 # small terms first would be more accurate; the MFMA order is large terms first
        acc = acc + pa[i] @ pb[j]
    return acc


FORMS = [
    ('fp32 (what runs today)', lambda a, b: a @ b),
    ('bf16, 1 product', lambda a, b: split_matmul(a, b, 1, [(0, 0)])),
    ('bf16 2-way, 3 products', lambda a, b: split_matmul(a, b, 2, [(0, 0), (0, 1), (1, 0)])),
    ('bf16 3-way, 6 products', lambda a, b: split_matmul(a, b, 3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)])),
    ('bf16 3-way, 9 products', lambda a, b: split_matmul(a, b, 3, [(i, j) for i in range(3) for j in range(3)])),
]


def report(name, a, b, post=None):
    """a [M, K], b [K, N] fp32.  post: optional function applied to the product before comparing (e.g. normalisation)."""
    ref = a.double() @ b.double()
    if post:
        ref = post(ref)
    scale = float(ref.abs().max())
    rows = []
    for label, f in FORMS:
        c = f(a, b).double()
        if post:
            c = post(c)
        d = (c - ref).abs()
        rows.append((label, float(d.max()) / scale, float(d.pow(2).mean().sqrt()) / scale))
    print('%s   [%d x %d] x [%d x %d], max |ref| %.3g' % (name, a.shape[0], a.shape[1], b.shape[0], b.shape[1], scale))
    base = rows[0][1]
    for label, mx, rms in rows:
        print('    %-26s max err / max|ref| %.3e   rms %.3e   (x%.1f of fp32)' % (label, mx, rms, mx / base))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--freq', type=int, default=32)
    ap.add_argument('--trained', default=None)
    args = ap.parse_args()
    from geobi_gnn_amd import network, meshgen
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = network.DualGNN().to(dev)
    if args.trained:
        net.load_state_dict(torch.load(args.trained, map_location='cpu', weights_only=True))
    os.environ['GEOBI_NET_EXECUTOR'] = '0'
    from geobi_gnn_amd import executor
    executor.ENABLED = False                       # module path: the hooks below see every layer
    cap = {}
    net.gnn_v.register_forward_hook(lambda m, i, o: cap.__setitem__('feat_v', o.detach().cpu()))
    net.gnn_f.register_forward_hook(lambda m, i, o: cap.__setitem__('feat_f', o.detach().cpu()))

    def pre(m, inp):
        x, g = inp[0], inp[1]
        cap['x_r1'] = x.detach().cpu()
        cap['rowptr'], cap['col'] = g.rowptr_out.cpu().long(), g.col_out.cpu().long()
    net.gnn_f.r_conv1.register_forward_pre_hook(pre)
    dv, df = meshgen.synthetic_dual_data(args.freq, 0.2, seed=200)
    with torch.no_grad():
        net((dv.to(dev), df.to(dev)))
    torch.cuda.synchronize()
    print('operands: noisy icosphere n=%d, %s weights' % (args.freq, 'trained (%s)' % args.trained if args.trained else 'random-init (bench)'))
    # ---- heads: out = leaky(x W1^T + b1) W2^T + b2
    for tag, feat, fc1, fc2, unit in (('vertex head', cap['feat_v'], net.fc_v1, net.fc_v2, False),
                                      ('normal head', cap['feat_f'], net.fc_f1, net.fc_f2, True)):
        w1, b1 = fc1.weight.detach().cpu(), fc1.bias.detach().cpu()
        w2, b2 = fc2.weight.detach().cpu(), fc2.bias.detach().cpu()
        report('%s, first GEMM x W1^T' % tag, feat, w1.t().contiguous())
        h64 = torch.nn.functional.leaky_relu(feat.double() @ w1.double().t() + b1.double(), 0.2)
        h = h64.float()
        post = (lambda o: torch.nn.functional.normalize(o + b2.double(), dim=1)) if unit else None
        report('%s, second GEMM h W2^T%s' % (tag, ' (then + b2, normalise: the unit normals)' if unit else ''), h,
               w2.t().contiguous(), post)
        # both GEMMs through the same form, end to end
        ref = h64 @ w2.double().t() + b2.double()
        if unit:
            ref = torch.nn.functional.normalize(ref, dim=1)
        print('%s, both GEMMs in one form, end to end:' % tag)
        base = None
        for label, f in FORMS:
            hh = torch.nn.functional.leaky_relu(f(feat, w1.t().contiguous()) + b1, 0.2)
            o = (f(hh, w2.t().contiguous()) + b2).double()
            if unit:
                o = torch.nn.functional.normalize(o, dim=1)
            e = float((o - ref).abs().max() / ref.abs().max())
            base = base or e
            extra = ''
            if unit:
                cosv = (o * ref).sum(1).clamp(-1, 1)
                extra = '   max angle to fp64 %.2e deg' % float(torch.rad2deg(torch.acos(cosv)).max())
            print('    %-26s max err / max|ref| %.3e   (x%.1f of fp32)%s' % (label, e, e / base, extra))
    # ---- node transform of gnn_f.r_conv1 (128 -> 64, level 1): out = z Wf, z_i[h, :] = mean_j q_ijh x_j in fp64
    x = cap['x_r1'].double()
    rowptr, col = cap['rowptr'], cap['col']
    conv = net.gnn_f.r_conv1
    N, C = x.shape
    u, c = conv.u.weight.detach().cpu().double(), conv.c.detach().cpu().double()
    lin = conv.lin.weight.detach().cpu()                      # [9 * 64, 128]
    deg = (rowptr[1:] - rowptr[:-1])
    row = torch.repeat_interleave(torch.arange(N), deg)
    p = x @ u.t()
    logit_e = p[col] - p[row] + c                             # u(x_j - x_i) + c per edge
    logit_s = c.expand(N, 9)                                  # self loop
    # the softmax runs over the 9 heads of ONE edge (FeaStConv), not over a node's edges
    qe = torch.softmax(logit_e, dim=1)
    qs = torch.softmax(logit_s, dim=1)
    z = torch.zeros(N, 9, C, dtype=torch.float64)
    z.index_add_(0, row, qe.unsqueeze(2) * x[col].unsqueeze(1))
    z += qs.unsqueeze(2) * x.unsqueeze(1)
    z /= (deg + 1).double().view(N, 1, 1)
    a = z.reshape(N, 9 * C).float()
    wf = lin.view(9, 64, C).permute(0, 2, 1).reshape(9 * C, 64).contiguous()       # row h * C + k = lin.weight[h * 64 + o, k]
    report('gnn_f.r_conv1 (128 -> 64, level 1, N = %d): node transform z Wf' % N, a, wf)
    g = torch.randn(N, 64)
    g *= 1e-3
    report('its backward dz = g Wf^T (g ~ 1e-3 N(0,1))', g, wf.t().contiguous())
    report('its weight gradient x^T r (r ~ 1e-3 N(0,1), K = N = %d)' % N, cap['x_r1'].t().contiguous(), torch.randn(N, 9 * 64) * 1e-3)


if __name__ == '__main__':
    main()
