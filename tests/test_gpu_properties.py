"""Size-independent properties of the FeaSt layer on the HIP path:
they hold for any weights and need no oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    return torch.device('cuda:0')


def _sym_graph(n, m, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (2, m), generator=g)
    ei = torch.cat([a, a.flip(0)], 1)
    key = torch.unique(ei[0] * n + ei[1])
    return torch.stack([key // n, key % n], 0)


def _conv(dev, cin, cout, seed):
    from geobi_gnn_amd.feast_conv import FeaStConv
    torch.manual_seed(seed)
    conv = FeaStConv(cin, cout, 9).to(dev)
    with torch.no_grad():                       # the default init zeroes c / bias: make them count
        conv.c.normal_()
        conv.bias.normal_()
        conv.u.weight.normal_(0, 0.5)
    return conv


@pytest.mark.parametrize('cin,cout', [(6, 32), (64, 32), (128, 64)])
def test_permutation_equivariance(dev, cin, cout):
    """Renumbering the nodes permutes the rows of the output (and of the input gradient)."""
    n = 1500
    ei = _sym_graph(n, 6000, seed=cin)
    conv = _conv(dev, cin, cout, seed=cout)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, generator=g)
    gout = torch.randn(n, cout, generator=g)
    perm = torch.randperm(n, generator=g)        # new id of old node i = perm[i]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n)

    xa = x.to(dev).requires_grad_(True)
    out = conv(xa, ei.to(dev))
    out.backward(gout.to(dev))

    xb = x[inv].to(dev).requires_grad_(True)     # row perm[i] holds old node i
    out_p = conv(xb, perm[ei].to(dev))
    out_p.backward(gout[inv].to(dev))

    scale = float(out.detach().abs().max())
    assert float((out_p[perm.to(dev)] - out).detach().abs().max()) <= 1e-5 * scale
    assert float((xb.grad[perm.to(dev)] - xa.grad).abs().max()) <= 1e-5 * float(xa.grad.abs().max())


def test_identical_heads_reduce_to_mean_aggregation(dev):
    """Soft-assignment weights sum to one: with W_h = W for every head the layer is
    out_i = mean_{j in N(i) + i} x_j W + bias, whatever u and c are."""
    n, cin, cout = 1200, 32, 64
    ei = _sym_graph(n, 5000, seed=3)
    conv = _conv(dev, cin, cout, seed=4)
    W = torch.randn(cout, cin)
    with torch.no_grad():
        conv.lin.weight.copy_(W.repeat(9, 1).to(dev))       # lin.weight [9*cout, cin], head-major
    x = torch.randn(n, cin)
    out = conv(x.to(dev), ei.to(dev)).detach().cpu()
    A = torch.zeros(n, n, dtype=torch.double)
    A[ei[1], ei[0]] = 1.0                                   # target <- source
    A.fill_diagonal_(1.0)                                   # existing loops dropped, exactly one re-added
    A /= A.sum(1, keepdim=True)
    ref = (A @ x.double()) @ W.double().t() + conv.bias.detach().cpu().double()
    assert float((out.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_constant_features(dev):
    """x_j = v for all j: every attention logit is c, so out_i = sum_h softmax(c)_h W_h v + bias on every node,
    isolated ones included."""
    n, cin, cout = 900, 12, 32
    ei = _sym_graph(n - 50, 3000, seed=5)                   # the last 50 nodes have no neighbours
    conv = _conv(dev, cin, cout, seed=6)
    v = torch.randn(cin)
    out = conv(v.repeat(n, 1).to(dev), ei.to(dev)).detach().cpu().double()
    q = torch.softmax(conv.c.detach().cpu().double(), 0)
    Wh = conv.lin.weight.detach().cpu().double().view(9, cout, cin)
    ref = torch.einsum('h,hoc,c->o', q, Wh, v.double()) + conv.bias.detach().cpu().double()
    assert float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_translation_leaves_attention_unchanged(dev):
    """q depends on x only through x_j - x_i: with a translation-blind lin (W_h t = 0) the output does not move."""
    n, cin, cout = 1000, 32, 32
    ei = _sym_graph(n, 4000, seed=7)
    conv = _conv(dev, cin, cout, seed=8)
    t = torch.randn(cin)
    t = t / t.norm()
    with torch.no_grad():
        W = conv.lin.weight.detach().cpu()
        conv.lin.weight.copy_((W - (W @ t)[:, None] * t[None, :]).to(dev))      # project t out of every row
    x = torch.randn(n, cin)
    a = conv(x.to(dev), ei.to(dev))
    b = conv((x + 3.0 * t).to(dev), ei.to(dev))
    assert float((a - b).detach().abs().max()) <= 2e-5 * float(a.detach().abs().max())
