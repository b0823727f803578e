"""Edge-sorted adjacency of one graph level, resident in HBM.

The reference passes COO ``edge_index [2, E]`` (int64; row = source j, col = target i) to every
FeaStConv / PoolingLayer call and lets PyG rebuild what it needs per call
(/root/reference/code/network.py:271-299, net_util.py:127).  Here a level's structure is built once
and cached on the ``edge_index`` tensor object:

  out-CSR  rowptr_out / col_out   per SOURCE node its targets, (row, col)-sorted -- the order the
                                  pooling kernels and pool_edge's output use
  in-CSR   rowptr_in / col_in     per TARGET node its sources -- what the FeaSt aggregation walks
  pos_in   [E]                    position of out-edge e inside the in-CSR (backward transposition)
  eid_out  [E]                    original COO column of out-edge e (level 0 only)

Self loops are never stored: FeaStConv re-adds exactly one per node (applied implicitly by the
kernels) and the pooling layer drops them first (net_util.py:163).
"""
import copy

import torch

from . import _lib as L


class Graph(object):
    def __init__(self, num_nodes, device):
        self.N = int(num_nodes)
        self.device = device
        self.E = 0
        self.rowptr_out = self.col_out = self.row_out = self.eid_out = None
        self.rowptr_in = self.col_in = self.pos_in = None
        self._coo64 = None
        self._view = None
        self._base = None          # set on the view attached to coo64(): shares the parent's arrays
        self._w_src = self._w_sorted = None
        self.symmetric = None      # True: in-CSR == out-CSR (checked at level 0, inherited by pooled graphs)

    # ------------------------------------------------------------------ construction
    @staticmethod
    def from_edge_index(edge_index, num_nodes):
        """COO int64 -> out-CSR (one host sync for the kept-edge count; cached per tensor)."""
        L.require_device(edge_index, 'edge_index')
        g = Graph(num_nodes, edge_index.device)
        ei = edge_index.contiguous()
        E0 = ei.shape[1]
        dev = ei.device
        g.rowptr_out = torch.empty(g.N + 1, dtype=torch.int32, device=dev)
        col = torch.empty(max(E0, 1), dtype=torch.int32, device=dev)
        eid = torch.empty(max(E0, 1), dtype=torch.int32, device=dev)
        nb = L.size_query('geobi_csr_ws_bytes', E0, g.N)
        ws = L.workspace(nb, dev)
        bad = torch.empty(1, dtype=torch.int32, device=dev)
        L.call('geobi_csr_from_coo', L.ptr(ei[0]), L.ptr(ei[1]), E0, g.N, 1, L.ptr(g.rowptr_out), L.ptr(col),
               L.ptr(eid), L.ptr(bad), L.ptr(ws), ws.numel(), L.stream())
        g.E = L.read_i32(g.rowptr_out[g.N:g.N + 1], 1)[0]
        nbad = L.read_i32(bad, 1)[0]
        if nbad:
            raise L.GeobiError('edge_index holds %d entries with node ids outside [0, %d)' % (nbad, g.N))
        g.col_out, g.eid_out = col[:g.E], eid[:g.E]
        g._check_symmetric()
        return g

    def _reverse_index(self, check=True):
        """pos[e] = position of the reverse of out-edge e; flag (check=True) != 0 if some edge has none."""
        dev = self.device
        pos = torch.empty(max(self.E, 1), dtype=torch.int32, device=dev)[:self.E]
        flag = torch.empty(1, dtype=torch.int32, device=dev) if check else None       # zeroed by the call
        L.call('geobi_csr_reverse_index', L.ptr(self.rowptr_out), L.ptr(self.ensure_rows()), L.ptr(self.col_out),
               self.E, L.ptr(pos), L.ptr(flag), L.stream())
        return pos, flag

    def _check_symmetric(self):
        """One extra host read at level-0 build time (cached with the graph)."""
        pos, flag = self._reverse_index()
        self.symmetric = int(flag.item()) == 0
        if self.symmetric:
            self.rowptr_in, self.col_in, self.pos_in = self.rowptr_out, self.col_out, pos

    @staticmethod
    def from_sorted(num_nodes, rowptr, row, col, symmetric=None):
        """Adopt pool_edge's (row, col)-sorted output as the out-CSR.  `symmetric` is inherited from
        the parent level: relabelling both endpoints of a symmetric edge set keeps it symmetric."""
        g = Graph(num_nodes, rowptr.device)
        g.rowptr_out, g.row_out, g.col_out = rowptr, row, col
        g.E = int(col.shape[0])
        g.symmetric = symmetric
        return g

    @staticmethod
    def union(graphs):
        """Disjoint union of loop-free sorted CSR graphs (node ids of graph k shifted by the node counts before
        it): the batched form of independent meshes / patches, built without going through COO."""
        dev = graphs[0].device
        g = Graph(sum(x.N for x in graphs), dev)
        rps, cols, noff, eoff = [], [], 0, 0
        for x in graphs:
            rps.append(x.rowptr_out[:x.N] + eoff)
            cols.append(x.col_out + noff)
            noff += x.N
            eoff += x.E
        rps.append(torch.full((1,), eoff, dtype=torch.int32, device=dev))
        g.rowptr_out, g.col_out, g.E = torch.cat(rps), torch.cat(cols), eoff
        g.symmetric = all(x.symmetric for x in graphs) if all(x.symmetric is not None for x in graphs) else None
        return g

    def ensure_rows(self):
        if self.row_out is None and self._base is not None:
            self.row_out = self._base.ensure_rows()
        if self.row_out is None:
            self.row_out = torch.empty(max(self.E, 1), dtype=torch.int32, device=self.device)[:self.E]
            L.call('geobi_expand_rowptr', L.ptr(self.rowptr_out), self.N, L.ptr(self.row_out), L.stream())
        return self.row_out

    def ensure_in(self):
        """Transposed CSR + edge correspondence (needed by the conv kernels)."""
        if self.rowptr_in is None and self._base is not None:
            b = self._base.ensure_in()
            self.rowptr_in, self.col_in, self.pos_in = b.rowptr_in, b.col_in, b.pos_in
        if self.rowptr_in is None and self.symmetric:
            pos, _ = self._reverse_index(check=False)      # symmetry is inherited: no flag, no memset
            self.rowptr_in, self.col_in, self.pos_in = self.rowptr_out, self.col_out, pos
        if self.rowptr_in is None:
            dev = self.device
            cap = max(self.E, 1)
            self.rowptr_in = torch.empty(self.N + 1, dtype=torch.int32, device=dev)
            self.col_in = torch.empty(cap, dtype=torch.int32, device=dev)[:self.E]
            pos_t = torch.empty(cap, dtype=torch.int32, device=dev)[:self.E]
            self.pos_in = torch.empty(cap, dtype=torch.int32, device=dev)[:self.E]
            ws = L.workspace(L.size_query('geobi_csr_ws_bytes', self.E, self.N), dev)
            L.call('geobi_csr_transpose', L.ptr(self.rowptr_out), L.ptr(self.col_out), self.N, self.E,
                   L.ptr(self.rowptr_in), L.ptr(self.col_in), L.ptr(pos_t), L.ptr(self.pos_in), L.ptr(ws),
                   ws.numel(), L.stream())
        return self

    # ------------------------------------------------------------------------ views
    def sorted_view(self):
        """This level seen through its (row, col)-sorted COO: same arrays, no input permutation."""
        if self.eid_out is None:
            return self
        if self._view is None:
            view = copy.copy(self)
            view.eid_out, view._base, view._w_src, view._w_sorted, view._coo64 = None, self, None, None, None
            view._view = view
            self._view = view
        return self._view

    def coo64(self):
        """Loop-free COO [2, E] int64 in (row, col)-sorted order -- what the module surface exposes."""
        v = self.sorted_view()
        if v._coo64 is None:
            v._coo64 = torch.stack([v.ensure_rows().long(), v.col_out.long()], 0)
            attach(v._coo64, v)
        return v._coo64

    def weights_sorted(self, edge_weight):
        """Original COO edge weights -> out-CSR order."""
        if self.eid_out is None:
            return edge_weight
        if edge_weight is self._w_src:
            return self._w_sorted
        out = torch.empty(max(self.E, 1), dtype=torch.float32, device=self.device)[:self.E]
        L.call('geobi_gather_f32', L.ptr(edge_weight.contiguous()), L.ptr(self.eid_out), self.E, L.ptr(out),
               L.stream())
        self._w_src, self._w_sorted = edge_weight, out
        return out


def attach(edge_index, graph):
    edge_index._geobi_graph = graph


def graph_of(edge_index, num_nodes):
    g = getattr(edge_index, '_geobi_graph', None)
    if g is None or g.N != int(num_nodes):
        g = Graph.from_edge_index(edge_index, num_nodes)
        attach(edge_index, g)
    return g
