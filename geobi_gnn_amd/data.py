"""Minimal graph container with the attribute surface the hot path touches.

The reference moves ``torch_geometric.data.Data`` objects through the network
(/root/reference/code/train_dual.py:201 ``d.to(device)``, :246 ``num_nodes``,
/root/reference/code/net_util.py:158 ``Data(x, edge_index, ..)``).  Only attribute
access, ``.to()``, ``num_nodes`` and ``hasattr`` semantics are needed on the path,
so this is a plain attribute bag -- no PyG dependency.
"""
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


def union_batch_graphs(dual_list):
    """``union_batch`` for device-resident pairs whose adjacency is already built (``Data.graph()``): the CSR
    graphs are concatenated directly (``Graph.union``) -- no COO round trip, no sort, no host sync -- and edge
    weights are taken in CSR order.  Same result fields as ``union_batch`` (``edge_index`` materialises
    lazily, loop-free)."""
    from .graph import Graph
    gvs = [dv.graph() for dv, _ in dual_list]
    gfs = [df.graph() for _, df in dual_list]
    ptr_v, ptr_f, fv = [0], [0], []
    for dv, df in dual_list:
        fv.append(df.fv_indices + ptr_v[-1])
        ptr_v.append(ptr_v[-1] + dv.x.shape[0])
        ptr_f.append(ptr_f[-1] + df.x.shape[0])

    def cat(items):
        return torch.cat(items) if all(t is not None for t in items) else None

    data_v = Data(torch.cat([dv.x for dv, _ in dual_list]), None, y=cat([dv.y for dv, _ in dual_list]),
                  depth_direction=cat([getattr(dv, 'depth_direction', None) for dv, _ in dual_list]), name='union-v')
    data_f = Data(torch.cat([df.x for _, df in dual_list]), None, y=cat([df.y for _, df in dual_list]),
                  fv_indices=torch.cat(fv), name='union-f')
    data_v.set_graph(Graph.union(gvs))
    data_f.set_graph(Graph.union(gfs))
    parts = [getattr(df.fv_indices, '_geobi_fv', None) for _, df in dual_list]
    if all(p is not None and p[2] == dv.x.shape[0] for p, (dv, _) in zip(parts, dual_list)):
        # every part's face table is already validated: so is the union (no range check = no host read later)
        from .network import mark_face_table
        fv32 = torch.cat([p[0] + off for p, off in zip(parts, ptr_v[:-1])])
        mark_face_table(data_f.fv_indices, fv32, ptr_v[-1])
    data_v.edge_weight = torch.cat([g.weights_sorted(dv.edge_weight) for g, (dv, _) in zip(gvs, dual_list)])
    data_f.edge_weight = torch.cat([g.weights_sorted(df.edge_weight) for g, (_, df) in zip(gfs, dual_list)])
    data_v.mesh_ptr = torch.tensor(ptr_v, dtype=torch.long)
    data_f.mesh_ptr = torch.tensor(ptr_f, dtype=torch.long)
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
