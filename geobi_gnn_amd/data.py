"""Minimal graph container with the attribute surface the hot path touches.

The reference moves ``torch_geometric.data.Data`` objects through the network
(/root/reference/code/train_dual.py:201 ``d.to(device)``, :246 ``num_nodes``,
/root/reference/code/net_util.py:158 ``Data(x, edge_index, ..)``).  Only attribute
access, ``.to()``, ``num_nodes`` and ``hasattr`` semantics are needed on the path,
so this is a plain attribute bag -- no PyG dependency.
"""
import ctypes

import torch


class Data(object):
    def __init__(self, x=None, edge_index=None, **kwargs):
        self.x = x
        self._graph = None
        self.edge_index = edge_index
        # the reference relies on these reading as None when absent
        # (net_util.py:81 ``data.pos``; :162 ``hasattr(data, 'edge_weight')``)
        self.edge_weight = None
        self.pos = None
        self.y = None
        for k, v in kwargs.items():
            setattr(self, k, v)

    # ``edge_index`` of a pooled level is materialised (int64 COO) only when somebody reads it; the
    # network itself walks the cached CSR (``graph``) and never needs the tensor.
    @property
    def edge_index(self):
        ei = self.__dict__.get('_edge_index')
        if ei is None and self._graph is not None:
            ei = self._graph.coo64()
            self.__dict__['_edge_index'] = ei
        return ei

    @edge_index.setter
    def edge_index(self, value):
        self.__dict__['_edge_index'] = value
        if value is not None:
            self._graph = getattr(value, '_geobi_graph', None)

    def graph(self, num_nodes=None):
        """Cached adjacency (geobi_gnn_amd.graph.Graph) of this level."""
        if self._graph is None:
            from .graph import graph_of
            n = self.num_nodes if num_nodes is None else num_nodes
            self._graph = graph_of(self.edge_index, n)
        return self._graph

    def set_graph(self, graph):
        """Adopt a level whose COO tensor has not been materialised."""
        self._graph = graph
        self.__dict__['_edge_index'] = graph._coo64

    def keys(self):
        ks = [k for k, v in self.__dict__.items() if v is not None and not k.startswith('_')]
        return ks + (['edge_index'] if self.edge_index is not None else [])

    @property
    def num_nodes(self):
        for k in ('x', 'pos', 'y'):
            v = getattr(self, k, None)
            if torch.is_tensor(v):
                return v.shape[0]
        if self._graph is not None:
            return self._graph.N
        if self.edge_index is not None and self.edge_index.numel() > 0:
            return int(self.edge_index.max()) + 1
        return 0

    @property
    def num_edges(self):
        return 0 if self.edge_index is None else self.edge_index.shape[1]

    def to(self, device, non_blocking=False):
        out = Data()
        for k, v in self.__dict__.items():
            if k == '_graph':
                continue
            if k == '_edge_index':
                k, v = 'edge_index', self.edge_index
            if torch.is_tensor(v):
                v = v.to(device, non_blocking=non_blocking)
            setattr(out, k, v)
        return out

    def shallow_copy(self):
        """New attribute bag over the SAME tensors (the forward pass rewrites .x / .edge_index on
        the bag it is given, like the reference does; cached graph structure stays attached to
        the shared edge_index tensor)."""
        out = Data()
        out.__dict__.update(self.__dict__)
        return out

    def clone(self):
        out = Data()
        for k, v in self.__dict__.items():
            if k == '_edge_index':
                k, v = 'edge_index', self.edge_index
            elif k.startswith('_'):
                continue
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out

    def __repr__(self):
        parts = []
        for k in self.keys():
            v = getattr(self, k)
            parts.append('%s=%s' % (k, list(v.shape) if torch.is_tensor(v) else repr(v)))
        return 'Data(%s)' % ', '.join(parts)


def union_batch(dual_list):
    """Disjoint union of several (data_v, data_f) pairs into one pair.

    Equivalent to the reference's gradient accumulation over ``batch_size``
    sequential single-mesh steps (/root/reference/code/train_dual.py:211-218) when the
    loss is averaged per mesh; pooling never links two components because graclus
    only matches along edges.  ``mesh_ptr_v`` / ``mesh_ptr_f`` keep the per-mesh node
    ranges so losses can be averaged per mesh.
    """
    xs_v, ys_v, ei_v, ew_v, dd_v = [], [], [], [], []
    xs_f, ys_f, ei_f, ew_f, fv = [], [], [], [], []
    ptr_v, ptr_f = [0], [0]
    for dv, df in dual_list:
        ov, of = ptr_v[-1], ptr_f[-1]
        xs_v.append(dv.x); ei_v.append(dv.edge_index + ov); ew_v.append(dv.edge_weight)
        xs_f.append(df.x); ei_f.append(df.edge_index + of); ew_f.append(df.edge_weight)
        fv.append(df.fv_indices + ov)
        if dv.y is not None:
            ys_v.append(dv.y)
        if df.y is not None:
            ys_f.append(df.y)
        if getattr(dv, 'depth_direction', None) is not None:
            dd_v.append(dv.depth_direction)
        ptr_v.append(ov + dv.x.shape[0])
        ptr_f.append(of + df.x.shape[0])
    # vertex graphs carry their self loops appended at the end (dataset.py:212-213);
    # the union keeps each mesh's block contiguous, which the kernels do not rely on.
    data_v = Data(torch.cat(xs_v), torch.cat(ei_v, 1), edge_weight=torch.cat(ew_v),
                  y=torch.cat(ys_v) if ys_v else None,
                  depth_direction=torch.cat(dd_v) if dd_v else None, name='union-v')
    data_f = Data(torch.cat(xs_f), torch.cat(ei_f, 1), edge_weight=torch.cat(ew_f),
                  y=torch.cat(ys_f) if ys_f else None, fv_indices=torch.cat(fv), name='union-f')
    data_v.mesh_ptr = torch.tensor(ptr_v, dtype=torch.long)
    data_f.mesh_ptr = torch.tensor(ptr_f, dtype=torch.long)
    return data_v, data_f


class _CopySeg(ctypes.Structure):
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('n', ctypes.c_int64), ('add', ctypes.c_int32),
                ('value', ctypes.c_float)]


class _Concat(object):
    """Collects the copy jobs of a union batch (geobi_concat32): every int32 array in one launch, every fp32 array in
    another, instead of a cat + add per array and mesh (~50 small launches and as many allocations per batch)."""

    def __init__(self, device):
        self.dev, self.jobs, self.keep = device, {0: [], 1: []}, []

    def cat(self, parts, adds=None, tail=None, shape=None):
        """parts: contiguous int32 / float32 tensors; adds: per-part int offsets (int32 only); tail: one closing int32."""
        if parts[0].dtype not in (torch.float32, torch.int32):
            raise ValueError('union batch: float32 / int32 arrays only (got %s): convert before the collate step' % parts[0].dtype)
        is_float = 1 if parts[0].dtype == torch.float32 else 0
        total = sum(t.numel() for t in parts) + (1 if tail is not None else 0)
        out = torch.empty(total, dtype=parts[0].dtype, device=self.dev)
        off = 0
        for k, t in enumerate(parts):
            if not t.is_contiguous() or t.dtype != parts[0].dtype or t.device != self.dev:
                raise ValueError('union batch: parts must be contiguous tensors of one dtype on %s' % self.dev)
            self.keep.append(t)
            self.jobs[is_float].append((t.data_ptr(), out.data_ptr() + 4 * off, t.numel(), 0 if adds is None else int(adds[k]), 0.0))
            off += t.numel()
        if tail is not None:
            self.jobs[0].append((None, out.data_ptr() + 4 * off, 1, int(tail), 0.0))
        return out if shape is None else out.view(shape)

    def fill(self, sizes, values):
        out = torch.empty(sum(sizes), dtype=torch.float32, device=self.dev)
        off = 0
        for n, v in zip(sizes, values):
            self.jobs[1].append((None, out.data_ptr() + 4 * off, n, 0, float(v)))
            off += n
        return out

    def run(self):
        from . import _lib as L
        for is_float in (0, 1):
            jobs = self.jobs[is_float]
            if jobs:
                arr = (_CopySeg * len(jobs))(*[_CopySeg(*j) for j in jobs])
                L.call('geobi_concat32', ctypes.cast(arr, ctypes.c_void_p), len(jobs), is_float, L.stream())
        self.keep = None


def union_batch_graphs(dual_list):
    """``union_batch`` for device-resident pairs whose adjacency is already built (``Data.graph()``): the collate step
    of a training loop (train_dual.py:199-201 hands the meshes over one by one; a batch is their disjoint union).  Every
    array of the union -- row pointers, neighbour ids, weights, features, targets, face table, and, where every part
    already has them, the reverse-edge index, the vertex -> corner lists and the per-mesh loss weights -- is an
    offset-shifted concatenation of the parts' arrays: TWO launches (geobi_concat32: int32 arrays, fp32 arrays), no
    sort, no host sync.  Same result fields as ``union_batch`` (``edge_index`` materialises lazily, loop-free)."""
    from .graph import Graph
    from .network import mark_face_table, _CornerIndex
    from . import ops
    dev = dual_list[0][0].x.device
    B = len(dual_list)
    gvs = [dv.graph() for dv, _ in dual_list]
    gfs = [df.graph() for _, df in dual_list]
    nv = [dv.x.shape[0] for dv, _ in dual_list]
    nf = [df.x.shape[0] for _, df in dual_list]
    off_v = [sum(nv[:k]) for k in range(B + 1)]
    off_f = [sum(nf[:k]) for k in range(B + 1)]
    cc = _Concat(dev)

    def graph_union(graphs, noff):
        g = Graph(noff[-1], dev)
        eoff = [sum(x.E for x in graphs[:k]) for k in range(B + 1)]
        g.E = eoff[-1]
        g.rowptr_out = cc.cat([x.rowptr_out[:x.N] for x in graphs], eoff[:-1], tail=eoff[-1])
        g.col_out = cc.cat([x.col_out for x in graphs], noff[:-1]) if g.E else torch.empty(0, dtype=torch.int32, device=dev)
        g.symmetric = all(x.symmetric for x in graphs) if all(x.symmetric is not None for x in graphs) else None
        if g.symmetric and g.E and all(x.pos_in is not None for x in graphs):
            # the parts' reverse-edge indices, shifted: the union needs no search of its own
            g.rowptr_in, g.col_in = g.rowptr_out, g.col_out
            g.pos_in = cc.cat([x.pos_in for x in graphs], eoff[:-1])
        return g

    def flt(items, cols=None):
        if any(t is None for t in items):
            return None
        out = cc.cat([t.contiguous() for t in items])
        return out if cols is None else out.view(-1, cols)

    g_v, g_f = graph_union(gvs, off_v), graph_union(gfs, off_f)
    data_v = Data(flt([dv.x for dv, _ in dual_list], dual_list[0][0].x.shape[1]), None,
                  y=flt([dv.y for dv, _ in dual_list], 3),
                  depth_direction=flt([getattr(dv, 'depth_direction', None) for dv, _ in dual_list], 3), name='union-v')
    data_f = Data(flt([df.x for _, df in dual_list], dual_list[0][1].x.shape[1]), None,
                  y=flt([df.y for _, df in dual_list], 3), name='union-f')
    data_v.set_graph(g_v)
    data_f.set_graph(g_f)
    data_v.edge_weight = flt([g.weights_sorted(dv.edge_weight) for g, (dv, _) in zip(gvs, dual_list)])
    data_f.edge_weight = flt([g.weights_sorted(df.edge_weight) for g, (_, df) in zip(gfs, dual_list)])
    # face table: int32 form shifted by the vertex offsets; validated parts make a validated union (no range check)
    parts = [getattr(df.fv_indices, '_geobi_fv', None) for _, df in dual_list]
    valid = all(p is not None and p[2] == n for p, n in zip(parts, nv))
    fv32_parts = [p[0] if valid else df.fv_indices.to(torch.int32).contiguous() for p, (_, df) in zip(parts, dual_list)]
    fv32 = cc.cat([t.view(-1) for t in fv32_parts], off_v[:-1]).view(-1, 3)
    corner = None
    if valid and all(p[1].index is not None for p in parts):
        # vertex -> corner lists of the parts (a corner id is 3 * face + slot), shifted: no sort for the union
        segptr = cc.cat([p[1].index.segptr[:n] for p, n in zip(parts, nv)], [3 * o for o in off_f[:-1]], tail=3 * off_f[-1])
        members = cc.cat([p[1].index.members for p in parts], [3 * o for o in off_f[:-1]])
        corner = (segptr, members)
    # per-row loss weights 1 / (B n_mesh): what parallel.batched_losses would build from mesh_ptr
    # (formed in fp32 exactly as parallel._mesh_weights forms them)
    per_mesh = lambda counts: (1.0 / (torch.tensor(counts, dtype=torch.float32) * B)).tolist()
    lw_v = cc.fill(nv, per_mesh(nv)) if B > 1 else None
    lw_f = cc.fill(nf, per_mesh(nf)) if B > 1 else None
    cc.run()
    data_f.fv_indices = fv32.long()
    if valid:
        mark_face_table(data_f.fv_indices, fv32, off_v[-1])
        if corner is not None:
            data_f.fv_indices._geobi_fv[1].index = ops.SegmentIndex.view(fv32.view(-1), off_v[-1], corner[0], corner[1])
    data_v.mesh_ptr = torch.tensor(off_v, dtype=torch.long)
    data_f.mesh_ptr = torch.tensor(off_f, dtype=torch.long)
    if B > 1:
        data_v._loss_weights, data_f._loss_weights = lw_v, lw_f
    return data_v, data_f


# ------------------------------------------------------------------- processed-file cache
# The reference caches every preprocessed (sub)mesh as ``torch.save((data_v, data_f), name.pt)`` of pickled
# PyG ``Data`` objects (/root/reference/code/dataset.py:153,182; read back at :276).  Unpickling needs PyG and
# executes code from the file; this cache holds the same fields as a plain {name: tensor | number | str}
# dict, so ``torch.load(..., weights_only=True)`` reads it.
_SCALARS = (int, float, str, bool)


def _data_to_dict(d):
    out = {}
    for k in d.keys():
        v = getattr(d, k)
        if torch.is_tensor(v):
            out[k] = v.detach().cpu()
        elif isinstance(v, _SCALARS):
            out[k] = v
        elif isinstance(v, dict) and k == 'meta':
            out[k] = {mk: (mv.detach().cpu() if torch.is_tensor(mv) else mv) for mk, mv in v.items()}
    return out


def _data_from_dict(dct):
    d = Data()
    for k, v in dct.items():
        setattr(d, k, v)
    return d


def save_processed(dual_data, path):
    """(data_v, data_f) -> one ``.pt`` file (tensors moved to the CPU; loop-free CSR-built graphs are stored
    through their materialised ``edge_index``)."""
    torch.save({'format': 'geobi-dual-v1', 'v': _data_to_dict(dual_data[0]), 'f': _data_to_dict(dual_data[1])}, path)


def load_processed(path, device=None):
    obj = torch.load(path, map_location='cpu', weights_only=True)
    if not isinstance(obj, dict) or obj.get('format') != 'geobi-dual-v1':
        raise ValueError('%s is not a geobi processed file' % path)
    dv, df = _data_from_dict(obj['v']), _data_from_dict(obj['f'])
    if device is not None:
        dv, df = dv.to(device), df.to(device)
    return dv, df


# ------------------------------------------------------------------- augmentation
class RandomRotate(object):
    """Training-time augmentation of /root/reference/code/dataset.py:39-69: one random rotation (about z by
    default, about all three axes with ``z_rotated=False``) applied to the position and normal columns of
    ``x``, to ``y`` and to ``pos`` / ``centroid`` / ``depth_direction`` of every graph of the pair, in place,
    on whatever device the tensors live.  ``rng``: optional ``numpy.random.Generator`` (the reference draws
    from numpy's global state)."""

    def __init__(self, z_rotated=True, rng=None):
        self.z_rotated = z_rotated
        self.rng = rng

    def matrix(self):
        import numpy as np
        u = self.rng.uniform(size=3) if self.rng is not None else np.random.uniform(size=3)
        a = u * 2 * np.pi
        rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
        return rz if self.z_rotated else rz @ (ry @ rx)

    def __call__(self, data):
        m = self.matrix()
        for d in data:
            r = torch.from_numpy(m).to(dtype=d.x.dtype, device=d.x.device)
            d.x[:, 0:3] = d.x[:, 0:3] @ r
            d.x[:, 3:6] = d.x[:, 3:6] @ r
            if d.y is not None:
                d.y[:, 0:3] = d.y[:, 0:3] @ r
            for k in ('pos', 'centroid', 'depth_direction'):
                v = getattr(d, k, None)
                if torch.is_tensor(v):
                    setattr(d, k, v @ r)
        return data
