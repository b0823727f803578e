"""Graph pooling / unpooling behind the reference's ``net_util`` surface.

Mirrors /root/reference/code/net_util.py: ``PoolingLayer`` (:56-245), ``pool_edge`` (:289-295),
``pool_face`` (:298-302), ``pooling`` (:305-343), ``pooling_pre`` (:346-366), ``pooling_run``
(:369-380) -- same names, arguments and side effects (``unpooling_indices`` kept on the module,
``data.edge_index`` / ``data.edge_weight`` rewritten loop-free).  The kernels are in
libgeobi_hip.so; edge lists come back (row, col)-sorted, a permutation of the reference's order.

Matching differs from torch_cluster by design: graclus visits nodes in a random order (CPU) or
runs randomised propose/respond rounds (CUDA) and is not reproducible; the HIP matching is the
deterministic greedy matching in descending edge-weight order (same objective, same
``cluster = min(u, v)`` labelling).  ``PoolingLayer.graclus_fn`` lets a caller supply cluster
vectors instead (parity tests replay the reference's recorded clusters through it).
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from . import ops
from .data import Data
from .graph import Graph, graph_of, attach

MATCH_ROUNDS = 8           # measured: 5-8 rounds converge on mesh graphs (icosphere n = 11..32); resumed if not
MATCH_ROUNDS_MAX = 2048


def _i32(t):
    return t.to(torch.int32).contiguous()


def hip_match(graph, weight_sorted, rounds=MATCH_ROUNDS, state=None, status=None):
    """Heavy-edge matching on the out-CSR.

    Returns (cluster int32 [N] with undecided nodes closed as singletons, status int32 [1] = nodes
    still undecided after `rounds` rounds, state int32 [N] to resume from)."""
    dev = graph.device
    init = state is None
    if init:
        state = torch.empty(graph.N, dtype=torch.int32, device=dev)
    cluster = torch.empty(graph.N, dtype=torch.int32, device=dev)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = L.workspace(L.size_query('geobi_match_ws_bytes', graph.N), dev)
    w = None if weight_sorted is None else weight_sorted.contiguous()
    L.call('geobi_match_heavy_edge', L.ptr(graph.rowptr_out), L.ptr(graph.col_out), L.ptr(w), graph.N, rounds,
           1 if init else 0, L.ptr(state), L.ptr(cluster), L.ptr(status), L.ptr(ws), ws.numel(), L.stream())
    return cluster, status, state


def hip_match_coarsen(graph, weight_sorted, counters, rounds=MATCH_ROUNDS, state=None):
    """Matching + dense relabel + inverse lists in one library call (geobi_match_coarsen).

    Returns (raw cluster int32 [N] (graclus ids), cnew int32 [N], SegmentIndex sized by N, state)."""
    dev = graph.device
    init = state is None
    if init:
        state = torch.empty(graph.N, dtype=torch.int32, device=dev)
    cluster = torch.empty(graph.N, dtype=torch.int32, device=dev)
    cnew = torch.empty(graph.N, dtype=torch.int32, device=dev)
    sidx = ops.SegmentIndex(cnew, graph.N, build=False)
    ws = L.workspace(L.size_query('geobi_match_coarsen_ws_bytes', graph.N), dev)
    w = None if weight_sorted is None else weight_sorted.contiguous()
    L.call('geobi_match_coarsen', L.ptr(graph.rowptr_out), L.ptr(graph.col_out), L.ptr(w), graph.N, rounds,
           1 if init else 0, L.ptr(state), L.ptr(cluster), L.ptr(cnew), L.ptr(sidx.segptr), L.ptr(sidx.members),
           L.ptr(counters), L.ptr(ws), ws.numel(), L.stream())
    return cluster, cnew, sidx, state


def relabel(cluster32, count=None, rep_is_self=False):
    """consecutive_cluster: dense ids; returns (cnew int32 [N], count int32 [1] on device).
    rep_is_self: every id is a member index with cluster[id] == id (a matching's min-member ids)."""
    n = cluster32.shape[0]
    dev = cluster32.device
    cnew = torch.empty(n, dtype=torch.int32, device=dev)
    if count is None:
        count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = L.workspace(L.size_query('geobi_relabel_ws_bytes', n), dev)
    L.call('geobi_relabel_compact', L.ptr(cluster32), n, 1 if rep_is_self else 0, L.ptr(cnew), L.ptr(count), L.ptr(ws),
           ws.numel(), L.stream())
    return cnew, count


def _pool_edge_raw(cnew32, graph, weight_sorted, count=None):
    """pool_edge on the out-CSR; worst-case sized outputs + device edge count."""
    dev = graph.device
    E, nmax = graph.E, graph.N
    cap = max(E, 1)
    rowptr_c = torch.empty(nmax + 1, dtype=torch.int32, device=dev)
    row_c = torch.empty(cap, dtype=torch.int32, device=dev)
    col_c = torch.empty(cap, dtype=torch.int32, device=dev)
    w_c = None if weight_sorted is None else torch.empty(cap, dtype=torch.float32, device=dev)
    if count is None:
        count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = L.workspace(L.size_query('geobi_pool_edge_ws_bytes', E), dev)
    L.call('geobi_pool_edge', L.ptr(cnew32), L.ptr(graph.ensure_rows()), L.ptr(graph.col_out),
           L.ptr(None if weight_sorted is None else weight_sorted.contiguous()), E, nmax, L.ptr(rowptr_c),
           L.ptr(row_c), L.ptr(col_c), L.ptr(w_c), L.ptr(count), L.ptr(ws), ws.numel(), L.stream())
    return rowptr_c, row_c, col_c, w_c, count


def _pool_edge_rows(cnew32, sidx, graph, weight_sorted, ncount, count, overflow):
    """Sort-free pool_edge for a matching (geobi_pool_edge_rows); worst-case sized outputs."""
    dev = graph.device
    cap = max(graph.E, 1)
    rowptr_c = torch.empty(graph.N + 1, dtype=torch.int32, device=dev)
    row_c = torch.empty(cap, dtype=torch.int32, device=dev)
    col_c = torch.empty(cap, dtype=torch.int32, device=dev)
    w_c = None if weight_sorted is None else torch.empty(cap, dtype=torch.float32, device=dev)
    ws = L.workspace(L.size_query('geobi_pool_edge_rows_ws_bytes', graph.N), dev)
    L.call('geobi_pool_edge_rows', L.ptr(cnew32), L.ptr(sidx.segptr), L.ptr(sidx.members), L.ptr(graph.rowptr_out),
           L.ptr(graph.col_out), L.ptr(None if weight_sorted is None else weight_sorted.contiguous()), L.ptr(ncount),
           graph.N, L.ptr(rowptr_c), L.ptr(row_c), L.ptr(col_c), L.ptr(w_c), L.ptr(count), L.ptr(overflow),
           L.ptr(ws), ws.numel(), L.stream())
    return rowptr_c, row_c, col_c, w_c


_FUSED_COARSEN = os.environ.get('GEOBI_FUSED_COARSEN', '1') == '1'
# GEOBI_SPIN_READ=0: read the pooling sizes with Tensor.tolist() (blocking copy) instead of geobi_read_i32
_SPIN_READ = os.environ.get('GEOBI_SPIN_READ', '1') == '1'


def _read4(counters):
    return L.read_i32(counters, 4) if _SPIN_READ else counters.tolist()


class _CounterPool(object):
    """Zeroed int32[4] scratch for the pooling kernels' device-side counters.  One fill kernel per 256 quads
    instead of one per pooling step: quads are handed out once and never reused; the backing tensor is
    released when its last quad is."""

    def __init__(self):
        self.buf, self.used, self.key = None, 0, None

    def take(self, device):
        key = (device.type, device.index)
        if self.buf is None or self.key != key or self.used >= self.buf.shape[0]:
            self.buf, self.used, self.key = torch.zeros((256, 4), dtype=torch.int32, device=device), 0, key
        q = self.buf[self.used]
        self.used += 1
        return q


    def take_block(self, k, device):
        """k consecutive zeroed quads as one [k, 4] tensor (one read-back serves all of them)."""
        key = (device.type, device.index)
        if self.buf is None or self.key != key or self.used + k > self.buf.shape[0]:
            self.buf, self.used, self.key = torch.zeros((256, 4), dtype=torch.int32, device=device), 0, key
        q = self.buf[self.used:self.used + k]
        self.used += k
        return q


_counters = _CounterPool()

# GEOBI_CHAIN_POOL=0: one size read-back per matching step (round-1 behaviour) instead of one per pooling layer
_CHAIN = os.environ.get('GEOBI_CHAIN_POOL', '1') == '1'


def _coarsen_chain(g0, w0, steps, rounds=MATCH_ROUNDS):
    """All `steps` matching steps of a pooling layer back to back, ONE host read at the end.

    Only the first step of a layer depends on features (through the edge weights it is handed); the later
    ones run on the pooled weights.  So the integer pipeline  match -> relabel -> lists -> pool_edge  of every
    step is enqueued without knowing the sizes of the previous one: step s + 1 runs on the coarse graph PADDED to
    the fine node count P -- rows past the true count are empty, i.e. isolated nodes that the matching closes as
    singletons with ids above every real cluster -- and the device-side counters of all steps come back in one
    copy.  Everything handed on is then cut to its exact size; nothing padded leaves this function, and the
    result is what the step-by-step path gives (the padding nodes never touch a real node's proposals).

    Returns None when a step overflowed the sort-free edge coarsening or did not converge in `rounds` rounds
    (the caller then takes the step-by-step path), else (clusts, raw, sidxs, coarse graph, coarse weights)."""
    P, dev = g0.N, g0.device
    ctr = _counters.take_block(steps, dev)
    g, w, outs = g0, w0, []
    for s in range(steps):
        cluster32, cnew, sidx, _ = hip_match_coarsen(g, w, ctr[s], rounds)
        rowptr_c, row_c, col_c, w_c = _pool_edge_rows(cnew, sidx, g, w, ctr[s][1:2], ctr[s][2:3], ctr[s][3:4])
        outs.append((cluster32, cnew, sidx, rowptr_c, row_c, col_c, w_c))
        if s + 1 < steps:
            g = Graph.from_sorted(P, rowptr_c, row_c, col_c, symmetric=g0.symmetric)     # padded to P rows
            w = w_c
    vals = L.read_i32(ctr, 4 * steps)
    real = [P]                                   # true node count per level
    edges = []
    for s in range(steps):
        undecided, nc_total, ec, overflow = vals[4 * s:4 * s + 4]
        if undecided or overflow:
            return None
        real.append(nc_total - (P - real[-1]))   # the padding nodes of the input came back as singletons
        edges.append(ec)
        if ec == 0:                              # the reference stops pooling once no edge is left (net_util.py:139)
            break
    clusts, raw, sidxs = [], [], []
    for s in range(len(edges)):
        cluster32, cnew, sidx, rowptr_c, row_c, col_c, w_c = outs[s]
        n_in, n_out = real[s], real[s + 1]
        exact = ops.SegmentIndex.view(cnew[:n_in], n_out, sidx.segptr[:n_out + 1], sidx.members[:n_in])
        clusts.append(exact.seg)
        raw.append(cluster32[:n_in])
        sidxs.append(exact)
    last = len(edges) - 1
    _, _, _, rowptr_c, row_c, col_c, w_c = outs[last]
    nc, ec = real[last + 1], edges[last]
    coarse = Graph.from_sorted(nc, rowptr_c[:nc + 1], row_c[:ec], col_c[:ec], symmetric=g0.symmetric)
    return clusts, raw, sidxs, coarse, (None if w_c is None else w_c[:ec])


def _coarsen(graph, weight_sorted, cluster32=None, rounds=MATCH_ROUNDS):
    """One pooling step on the structure: match (unless given) -> relabel -> pool_edge.

    One host sync reads {undecided nodes, coarse node count, coarse edge count, overflow}.
    Returns (cnew int32, coarse Graph, coarse weights, raw cluster int32, inverse lists or None)."""
    sidx = None
    if cluster32 is None:
        state, total = None, 0
        while True:
            # one int32[4] holds {undecided, N', E', overflow}: one fill, one device-to-host read per step
            counters = _counters.take(graph.device)
            if _FUSED_COARSEN:
                cluster32, cnew, sidx, state = hip_match_coarsen(graph, weight_sorted, counters, rounds, state)
            else:
                cluster32, _, state = hip_match(graph, weight_sorted, rounds, state, status=counters[0:1])
                cnew, _ = relabel(cluster32, count=counters[1:2], rep_is_self=True)
                sidx = ops.SegmentIndex.from_matching(cnew, cluster32, graph.N)
            total += rounds
            rowptr_c, row_c, col_c, w_c = _pool_edge_rows(cnew, sidx, graph, weight_sorted, counters[1:2],
                                                          counters[2:3], counters[3:4])
            undecided, nc, ec, overflow = _read4(counters)
            if overflow:     # a coarse node gathers > 64 fine entries: take the radix-sort path
                count = torch.zeros(1, dtype=torch.int32, device=graph.device)
                rowptr_c, row_c, col_c, w_c, _ = _pool_edge_raw(cnew, graph, weight_sorted, count=count)
                ec = int(count.item())
            # rare: proposal chains longer than the rounds run so far -> resume from the saved state.
            # Beyond MATCH_ROUNDS_MAX the undecided nodes stay singletons (still a valid clustering).
            if not undecided or total >= MATCH_ROUNDS_MAX:
                break
            rounds = min(rounds * 2, MATCH_ROUNDS_MAX - total)
        sidx.narrow(nc)
    else:
        counters = _counters.take(graph.device)
        cnew, _ = relabel(cluster32, count=counters[1:2])
        rowptr_c, row_c, col_c, w_c, _ = _pool_edge_raw(cnew, graph, weight_sorted, count=counters[2:3])
        _, nc, ec, _ = _read4(counters)
    coarse = Graph.from_sorted(nc, rowptr_c[:nc + 1], row_c[:ec], col_c[:ec], symmetric=graph.symmetric)
    return cnew, coarse, (None if w_c is None else w_c[:ec]), cluster32, sidx


def _pool_features(x, sidx, pool_type):
    return ops.apply_op(ops.SegmentMaxFn if pool_type == 'max' else ops.SegmentMeanFn, x, sidx)


def _compose(clusts):
    clust = clusts[-1]
    for c in clusts[-2::-1]:
        clust = clust[c]          # int32 index tensors are valid indices
    return clust


def _feature_gauss(x, graph, denom):
    """exp(-|x_i - x_j|^2 / denom) per sorted edge (types 1, 2, 8, 9); type 10 has its own kernel."""
    out = torch.empty(max(graph.E, 1), dtype=torch.float32, device=x.device)[:graph.E]
    L.call('geobi_edge_weight_t10', L.ptr(x.detach().contiguous()), x.shape[1], L.ptr(graph.ensure_rows()),
           L.ptr(graph.col_out), None, graph.E, L.ptr(out), L.stream())
    return out if denom == 2 else out.pow(2.0 / denom)


def _minmax(v):
    return (v - v.min()) / (v.max() - v.min() + 1e-12)


class PoolingLayer(nn.Module):
    def __init__(self, in_channel, pool_type='max', pool_step=2, edge_weight_type=0, wei_param=2):
        super().__init__()
        assert pool_type in ['max', 'mean']
        self.pool_type = pool_type
        self.pool_step = pool_step
        self.edge_weight_type = edge_weight_type
        self.wei_param = wei_param
        if self.edge_weight_type in [4, 5]:
            self.lin = nn.Linear(in_channel, in_channel)
        if self.edge_weight_type in [3, 4, 5]:
            self.att_l = nn.Parameter(torch.empty(1, in_channel))
            self.att_r = nn.Parameter(torch.empty(1, in_channel))
            nn.init.xavier_uniform_(self.att_l.data, gain=1.414)
            nn.init.xavier_uniform_(self.att_r.data, gain=1.414)
        self._unpool32 = self._unpool64 = None
        self.graclus_fn = None          # optional: callable(edge_index, weight, num_nodes) -> cluster
        self._last_clusters32 = None    # raw cluster vectors of the last forward
        self._unpool_index = None
        self._arena_guard = None        # set by the whole-network executor: state lives in a shared arena

    def _fresh(self):
        """After an inference pass through the whole-network executor the state below is a set of views into the
        device's shared arena: the next pass overwrites it, and a read after that must fail, not return another
        pass' numbers."""
        from . import executor
        if not executor.is_current(self._arena_guard):
            raise L.GeobiError('PoolingLayer state (unpooling_indices / last_clusters) of an earlier inference pass '
                               'was overwritten by a later forward on this device: read it before the next pass')

    @property
    def unpooling_indices(self):
        """Composed fine -> coarse index of the last forward (net_util.py:152-156), int64 like the reference
        keeps it; converted from the int32 the kernels use on first read."""
        self._fresh()
        if self._unpool64 is None and self._unpool32 is not None:
            self._unpool64 = self._unpool32.long()
        return self._unpool64

    @unpooling_indices.setter
    def unpooling_indices(self, value):
        self._arena_guard = None
        self._unpool64 = value
        self._unpool32 = None if value is None else value.to(torch.int32)

    @property
    def last_clusters(self):
        """Raw (pre-relabel) cluster vectors of the last forward, int64 like graclus returns them."""
        self._fresh()
        return None if self._last_clusters32 is None else [c.long() for c in self._last_clusters32]

    # -- edge weight fed to the matching (no gradient is needed: it only drives integer matching)
    def _get_edge_weight(self, data):
        x = data.x
        g = data.graph(x.shape[0])
        if g.E == 0:
            return None
        w = getattr(data, 'edge_weight', None)
        if w is not None:
            w = g.weights_sorted(w)
        # the reference rewrites its input loop-free (net_util.py:166-167): same here, with the
        # (row, col)-sorted COO materialised lazily on first read
        data.set_graph(g.sorted_view())
        data.edge_weight = w
        t = self.edge_weight_type
        if t == -1:
            return None
        if t == 0:
            return w
        if t == 10:
            out = torch.empty_like(w)
            L.call('geobi_edge_weight_t10', L.ptr(x.detach().contiguous()), x.shape[1], L.ptr(g.ensure_rows()),
                   L.ptr(g.col_out), L.ptr(w), g.E, L.ptr(out), L.stream())
            return out
        if t == 1:
            return _feature_gauss(x, g, self.wei_param)
        if t == 2:
            return w * _feature_gauss(x, g, self.wei_param)
        if t in (3, 4, 5):
            # learned (GAT-style) weights on the device: the Linear + leaky-relu of types 4 / 5 as one MFMA GEMM with
            # its epilogue, the node dots and the per-edge sigmoid as geobi_edge_weight_att
            xx = x.detach().contiguous()
            n, c = xx.shape
            if t != 3:
                lw, lb = self.lin.weight.detach().contiguous(), self.lin.bias.detach().contiguous()
                h = torch.empty(n, lw.shape[0], dtype=torch.float32, device=xx.device)
                L.call('geobi_gemm_nn', L.ptr(xx), c, L.ptr(lw), c, 1, L.ptr(h), lw.shape[0], n, lw.shape[0], c, L.ptr(lb),
                       0.2, L.stream())
                xx, c = h, lw.shape[0]
            if t == 5 and w is None:
                raise L.GeobiError('edge_weight_type 5 averages with data.edge_weight, which is missing')
            out = torch.empty(max(g.E, 1), dtype=torch.float32, device=xx.device)[:g.E]
            ws = torch.empty(2 * n, dtype=torch.float32, device=xx.device)
            L.call('geobi_edge_weight_att', L.ptr(xx), c, L.ptr(self.att_l.detach().reshape(-1).contiguous()),
                   L.ptr(self.att_r.detach().reshape(-1).contiguous()), L.ptr(g.ensure_rows()), L.ptr(g.col_out),
                   L.ptr(w) if t == 5 else None, n, g.E, L.ptr(ws), L.ptr(out), L.stream())
            return out
        if t == 6:
            return _minmax(w)
        if t == 7:
            return _minmax(torch.log(_feature_gauss(x, g, 2).clamp_min(1e-38)) * 2)
        if t == 8:
            return _minmax(_feature_gauss(x, g, 2))
        if t == 9:
            return _minmax(w) + _minmax(_feature_gauss(x, g, 2))
        return w

    def forward(self, data, visual=False):
        L.require_device(data.x, 'data.x')
        edge_weight = self._get_edge_weight(data)
        x, pos = data.x, getattr(data, 'pos', None)
        g = data.graph(x.shape[0])
        edge_dual = getattr(data, 'edge_dual', None)
        face = getattr(data, 'fv_indices', None)

        clusts, raw, sidxs = [], [], []
        chain = None
        if _CHAIN and self.graclus_fn is None and self.pool_step > 1 and g.E > 0 and g.N > 0:
            chain = _coarsen_chain(g, edge_weight, self.pool_step)
        if chain is not None:
            # structure of every step from one read-back; the features follow with exact sizes
            clusts, raw, sidxs, g, edge_weight = chain
            for cnew, sidx in zip(clusts, sidxs):
                x = _pool_features(x, sidx, self.pool_type)
                pos = None if pos is None else ops.apply_op(ops.SegmentMeanFn, pos, sidx)
                edge_dual = None if edge_dual is None else cnew.long()[edge_dual]
        else:
            for _ in range(self.pool_step):
                given = None
                if self.graclus_fn is not None:
                    given = _i32(self.graclus_fn(g.coo64(), edge_weight, g.N))
                cnew, g_c, w_c, cl_raw, sidx = _coarsen(g, edge_weight, given)
                raw.append(cl_raw)
                clusts.append(cnew)
                if sidx is None:
                    sidx = ops.SegmentIndex.from_matching(cnew, cl_raw, g_c.N)
                sidxs.append(sidx)
                x = _pool_features(x, sidx, self.pool_type)
                pos = None if pos is None else ops.apply_op(ops.SegmentMeanFn, pos, sidx)
                edge_dual = None if edge_dual is None else cnew.long()[edge_dual]
                g, edge_weight = g_c, w_c
                if g.E == 0:
                    break

        clust = _compose(clusts)
        self._arena_guard = None
        self._unpool32, self._unpool64 = clust, None
        uidx = sidxs[0]
        for nxt in sidxs[1:]:
            uidx = ops.SegmentIndex.compose(uidx, nxt, clust)
        self._unpool_index = uidx
        self._last_clusters32 = raw
        out = Data(x, None, edge_dual=edge_dual, edge_weight=edge_weight, pos=pos, fv_indices=face)
        out.set_graph(g)            # edge_index (int64 COO) materialises on first read
        return out

    def unpooling(self, x):
        if self._unpool32 is None:
            return x
        self._fresh()
        if self._unpool_index is None:       # forward ran through the whole-network executor: lists built on demand
            nseg = int(self._unpool32.max().item()) + 1
            self._unpool_index = ops.SegmentIndex(self._unpool32.contiguous(), nseg)
        return ops.apply_op(ops.UnpoolFn, x, self._unpool_index)


# ------------------------------------------------------------------------ functional API
def pool_edge(cluster, edge_index, edge_attr=None, op='mean'):
    """net_util.py:289-295.  cluster: consecutive ids [N]; returns ((row, col)-sorted COO, mean weights)."""
    if op != 'mean':
        raise NotImplementedError("pool_edge: only op='mean' is used by the reference")
    g = graph_of(edge_index, cluster.size(0))
    w = None if edge_attr is None else g.weights_sorted(edge_attr)
    rowptr_c, row_c, col_c, w_c, count = _pool_edge_raw(_i32(cluster), g, w)
    ec = int(count.item())
    ei = torch.stack([row_c[:ec].long(), col_c[:ec].long()], 0)
    return ei, (None if w_c is None else w_c[:ec])


def pool_face(cluster, fv_indices):
    face = cluster[fv_indices.view(-1)].view(-1, 3)
    invalid = (face[:, 0] == face[:, 1]) | (face[:, 0] == face[:, 2]) | (face[:, 1] == face[:, 2])
    return face[~invalid]


def pooling(data, p_type='max', level=2, wei_type=0, graclus_fn=None):
    """net_util.py:305-343: on-the-fly pooling; returns (coarse Data, composed cluster index).
    graclus_fn (not in the reference): callable(edge_index, weight, num_nodes) -> cluster, used instead of the
    built-in matching (parity tests replay the oracle's clusters through it, like PoolingLayer.graclus_fn)."""
    x, pos = data.x, getattr(data, 'pos', None)
    g = graph_of(data.edge_index, x.shape[0])
    if wei_type == 0:
        w = getattr(data, 'edge_weight', None)
        w = None if w is None else g.weights_sorted(w)
    elif wei_type == 1:
        nrm = (x.detach() ** 2).sum(1)
        w = ((nrm[g.ensure_rows().long()] - nrm[g.col_out.long()]) ** 2 / (-2)).exp()
    else:
        w = _feature_gauss(x, g, 2)
    clusts = []
    for _ in range(level):
        given = None if graclus_fn is None else _i32(graclus_fn(g.coo64(), w, g.N))
        cnew, g_c, w_c, cl_raw, sidx = _coarsen(g, w, given)
        clusts.append(cnew)
        if sidx is None:          # injected clusters need not be a matching: general inverse lists
            sidx = ops.SegmentIndex(cnew, g_c.N)
        x = _pool_features(x, sidx, p_type)
        pos = None if pos is None else ops.apply_op(ops.SegmentMeanFn, pos, sidx)
        g, w = g_c, w_c
        if g.E == 0:
            break
    return Data(x, g.coo64(), pos=pos, edge_weight=w), _compose(clusts).long()


def pooling_pre(data, step=2, level=2, graclus_fn=None):
    """net_util.py:346-366: precompute the cluster hierarchy from the static edge weights
    (graclus_fn: see `pooling`)."""
    n = data.num_nodes
    g = graph_of(data.edge_index, n)
    w = getattr(data, 'edge_weight', None)
    w = None if w is None else g.weights_sorted(w)
    for i in range(1, level + 1):
        clusters = []
        for _ in range(step):
            given = None if graclus_fn is None else _i32(graclus_fn(g.coo64(), w, g.N))
            cnew, g, w, _, _ = _coarsen(g, w, given)
            clusters.append(cnew.long())
        setattr(data, 'pool_l%d' % i, {'clusters': clusters, 'cluster_inv': _compose(clusters).long()})
    data.edge_weight = None
    return data


def pooling_run(data, pool_info, p_type='max'):
    """net_util.py:369-380: replay precomputed clusters."""
    x, pos = data.x, getattr(data, 'pos', None)
    g = graph_of(data.edge_index, x.shape[0])
    for clust in pool_info['clusters']:
        c32 = _i32(clust)
        nc = int(clust.max().item()) + 1
        sidx = ops.SegmentIndex(c32, nc)
        x = _pool_features(x, sidx, p_type)
        pos = None if pos is None else ops.apply_op(ops.SegmentMeanFn, pos, sidx)
        rowptr_c, row_c, col_c, _, count = _pool_edge_raw(c32, g, None)
        ec = int(count.item())
        g = Graph.from_sorted(nc, rowptr_c[:nc + 1], row_c[:ec], col_c[:ec])
    return Data(x, g.coo64(), pos=pos)
