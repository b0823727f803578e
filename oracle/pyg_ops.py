"""TEST INFRASTRUCTURE -- CPU restatement of the third-party primitives on the hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

The arithmetic of the reference's hot path lives in un-vendored, un-pinned wheels that
are absent here: torch_geometric (FeaStConv, graclus wrapper,
consecutive_cluster, pool_pos, self-loop utils, Data), torch_scatter (scatter),
torch_sparse (coalesce) and torch_cluster (graclus kernel).  Their published algorithms
are restated below in plain PyTorch, *keeping the reference's op decomposition* (per-edge
Linear layers, materialised x_i/x_j, index_add based scatter) so the CPU baseline pays
the same FLOPs and bytes as the reference would.

PARITY UNPINNED for these primitives: the reference holds no tests, golden vectors or
fixtures for them and the wheels cannot be imported, so they are anchored only on the
reference's call sites (cited per function) and on the algorithms' published definitions.
"""
import math
import ctypes
import os

import torch
import torch.nn.functional as F
from torch import nn


# ----------------------------------------------------------------------------- Data
class Data(object):
    """Attribute bag standing in for torch_geometric.data.Data (net_util.py:158 call shape)."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        self.x, self.edge_index, self.edge_attr, self.y, self.pos = x, edge_index, edge_attr, y, pos
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        for k in ('x', 'pos', 'y'):
            v = getattr(self, k, None)
            if torch.is_tensor(v):
                return v.shape[0]
        return int(self.edge_index.max()) + 1

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


# ------------------------------------------------------------------------ edge utils
def remove_self_loops(edge_index, edge_attr=None):
    """PyG utils: keep columns with row != col (net_util.py:163,292 call sites)."""
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    return edge_index, (None if edge_attr is None else edge_attr[mask])


def add_self_loops(edge_index, edge_attr=None, fill_value=1.0, num_nodes=None):
    """PyG utils: append (i, i) for every node at the END of the list (dataset.py:213)."""
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    out = torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], 1)
    if edge_attr is not None:
        edge_attr = torch.cat([edge_attr, edge_attr.new_full((n,) + edge_attr.shape[1:], fill_value)], 0)
    return out, edge_attr


def coalesce(index, value, m, n, op='add'):
    """torch_sparse.coalesce: sort by row*n+col, merge duplicates with ``op`` (net_util.py:294)."""
    key = index[0] * n + index[1]
    key_sorted, perm = torch.sort(key, stable=True)
    uniq, inv = torch.unique_consecutive(key_sorted, return_inverse=True)
    out_index = torch.stack([torch.div(uniq, n, rounding_mode='floor'), uniq % n], 0)
    if value is None:
        return out_index, None
    value = value[perm]
    red = {'add': 'sum', 'sum': 'sum', 'mean': 'mean', 'max': 'max', 'min': 'min'}[op]
    return out_index, scatter(value, inv, dim=0, dim_size=uniq.numel(), reduce=red)


def to_undirected(edge_index, num_nodes=None):
    """PyG utils: symmetrise + coalesce (dataset.py:212)."""
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    row, col = edge_index
    ei = torch.stack([torch.cat([row, col]), torch.cat([col, row])], 0)
    return coalesce(ei, None, n, n)[0]


# --------------------------------------------------------------------------- scatter
class _ScatterMax(torch.autograd.Function):
    """torch_scatter max: returns the segment max, gradient routed to ONE arg-max row."""

    @staticmethod
    def forward(ctx, src, index, dim_size):
        C = src.shape[1:]
        idx = index.view(-1, *([1] * len(C))).expand_as(src)
        out = src.new_zeros((dim_size,) + tuple(C))
        out = out.scatter_reduce(0, idx, src, 'amax', include_self=False)
        # first source row attaining the max (torch_scatter CPU: strict '>' keeps the first)
        pos = torch.arange(src.shape[0], device=src.device).view(-1, *([1] * len(C))).expand_as(src)
        big = src.shape[0]
        cand = torch.where(src == out[index], pos, torch.full_like(pos, big))
        arg = torch.full((dim_size,) + tuple(C), big, dtype=torch.long, device=src.device)
        arg = arg.scatter_reduce(0, idx, cand, 'amin', include_self=True)
        ctx.save_for_backward(arg)
        ctx.n_src = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad):
        (arg,) = ctx.saved_tensors
        g = grad.new_zeros((ctx.n_src + 1,) + tuple(grad.shape[1:]))
        g.scatter_(0, arg, grad)
        return g[:-1], None, None


def scatter(src, index, dim=0, dim_size=None, reduce='sum'):
    """torch_scatter.scatter along dim 0 (net_util.py:131-134; network.py:350)."""
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    if reduce in ('sum', 'add', 'mean'):
        out = src.new_zeros((dim_size,) + tuple(src.shape[1:])).index_add_(0, index, src)
        if reduce == 'mean':
            cnt = torch.bincount(index, minlength=dim_size).clamp(min=1).to(src.dtype)
            out = out / cnt.view(-1, *([1] * (src.dim() - 1)))
        return out
    if reduce == 'max':
        s = src if src.dim() > 1 else src.unsqueeze(1)
        out = _ScatterMax.apply(s, index, dim_size)
        return out if src.dim() > 1 else out.squeeze(1)
    if reduce == 'min':
        return -scatter(-src, index, dim, dim_size, 'max')
    raise ValueError(reduce)


# --------------------------------------------------------------------------- graclus
_GRACLUS_C = None


def _load_graclus_c():
    global _GRACLUS_C
    if _GRACLUS_C is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'liboracle_c.so')
        _GRACLUS_C = ctypes.CDLL(path) if os.path.exists(path) else False
    return _GRACLUS_C


def graclus_csr(edge_index, weight, num_nodes):
    """Shared preamble of torch_cluster graclus_cpu: drop loops, sort by row, rowptr."""
    row, col = edge_index
    mask = row != col
    row, col = row[mask], col[mask]
    w = None if weight is None else weight[mask]
    perm = torch.argsort(row, stable=True)
    row, col = row[perm], col[perm]
    w = None if w is None else w[perm]
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=num_nodes), 0)
    return rowptr, col, w


def graclus(edge_index, weight=None, num_nodes=None, node_perm=None, generator=None):
    """Greedy heavy-edge matching, torch_cluster graclus_cpu semantics (net_util.py:127).

    Nodes are visited in ``randperm`` order; an unmatched node u takes the unmatched
    neighbour with the largest weight (``>=`` so the last maximum wins, threshold 0), both
    get cluster id min(u, v); a node with no free neighbour keeps its own id.
    """
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    dev = edge_index.device
    rowptr, col, w = graclus_csr(edge_index.cpu(), None if weight is None else weight.detach().cpu(), n)
    if node_perm is None:
        node_perm = torch.randperm(n, generator=generator)
    out = torch.full((n,), -1, dtype=torch.long)
    lib = _load_graclus_c()
    if lib and w is not None and w.dtype == torch.float32:
        rp, cl, ww, pm = rowptr.contiguous(), col.contiguous(), w.contiguous(), node_perm.contiguous()
        lib.oracle_graclus_f32(ctypes.c_int64(n), ctypes.c_void_p(rp.data_ptr()), ctypes.c_void_p(cl.data_ptr()),
                               ctypes.c_void_p(ww.data_ptr()), ctypes.c_void_p(pm.data_ptr()),
                               ctypes.c_void_p(out.data_ptr()))
        return out.to(dev)
    rp, cl, pm = rowptr.tolist(), col.tolist(), node_perm.tolist()
    ww = None if w is None else w.tolist()
    o = [-1] * n
    for u in pm:
        if o[u] >= 0:
            continue
        v_max, w_max = u, 0.0
        for e in range(rp[u], rp[u + 1]):
            v = cl[e]
            if o[v] >= 0:
                continue
            if ww is None:
                v_max = v
                break
            if ww[e] >= w_max:
                v_max, w_max = v, ww[e]
        o[u] = o[v_max] = min(u, v_max)
    return torch.tensor(o, dtype=torch.long, device=dev)


def consecutive_cluster(src):
    """PyG pool.consecutive: dense relabel by sorted unique id; perm = last member per cluster."""
    unique, inv = torch.unique(src, sorted=True, return_inverse=True)
    perm = torch.arange(inv.size(0), dtype=inv.dtype, device=inv.device)
    perm = inv.new_empty(unique.size(0)).scatter_(0, inv, perm)
    return inv, perm


def pool_pos(cluster, pos):
    return scatter(pos, cluster, dim=0, reduce='mean')


# -------------------------------------------------------------------------- FeaStConv
class FeaStConv(nn.Module):
    """torch_geometric.nn.FeaStConv (PyG >= 2.0 parameter naming), per-edge op decomposition.

    Called 16x by the reference (network.py:258-268, 271-299).  q = softmax(u(x_j - x_i) + c);
    message = sum_h q_h * (W_h x_j); mean over the target node; + bias.  Self loops are
    removed and re-added (appended) before propagation.  Source j = edge_index[0],
    target i = edge_index[1].
    """

    def __init__(self, in_channels, out_channels, heads=1, add_self_loops=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.add_self_loops = add_self_loops
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.u = nn.Linear(in_channels, heads, bias=False)
        self.c = nn.Parameter(torch.empty(heads))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels)       # PyG Linear weight_initializer='uniform'
        nn.init.uniform_(self.lin.weight, -bound, bound)
        nn.init.uniform_(self.u.weight, -bound, bound)
        nn.init.normal_(self.c, mean=0.0, std=0.1)
        if self.bias is not None:
            nn.init.normal_(self.bias, mean=0.0, std=0.1)

    def forward(self, x, edge_index):
        n = x.shape[0]
        if self.add_self_loops:
            edge_index, _ = remove_self_loops(edge_index)
            edge_index, _ = add_self_loops(edge_index, num_nodes=n)
        x_j = x.index_select(0, edge_index[0])
        x_i = x.index_select(0, edge_index[1])
        q = F.softmax(self.u(x_j - x_i) + self.c, dim=1)
        m = self.lin(x_j).view(x_j.size(0), self.heads, -1)
        m = (m * q.view(-1, self.heads, 1)).sum(dim=1)
        out = scatter(m, edge_index[1], dim=0, dim_size=n, reduce='mean')
        if self.bias is not None:
            out = out + self.bias
        return out
