"""Mesh -> the hot path's inputs, on the MI355X.

Device counterpart of ``meshgen.build_dual_data`` -- i.e. of what the reference's loader computes on
the CPU with openmesh + PyG before the network runs:

* /root/reference/code/dataset.py:197-233  ``process_one_submesh`` (normals, vertex graph
  ``to_undirected(ev.T)`` + self loops, facet graph, bilateral weights, pos / normal per node),
* /root/reference/code/data_util.py:383-398 ``calc_weight``, :436-456 ``build_facet_graph``,
  :201-230 ``center_and_scale`` (s_type 0),
* /root/reference/code/dataset.py:246-269  ``post_processing`` (feature assembly).

The graphs are emitted directly as the (row, col)-sorted, loop-free CSR the conv / pooling kernels walk
(``graph.Graph``) -- no COO round trip, no sort.  ``edge_weight`` is stored in that order.  The int64 COO
``edge_index`` the module surface exposes is materialised lazily (loop-free); pass
``reference_layout=True`` to get the reference's exact tensors instead (vertex graph: sorted pairs then
V appended self loops; facet graph: self loops inline; loop weights as calc_weight gives them).
"""
import numpy as np
import torch

from . import _lib as L
from .data import Data
from .graph import Graph, attach


def _dev_i32(a, device):
    t = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a)
    return t.to(device=device, dtype=torch.int32).contiguous()


def vertex_faces(faces, num_vertices):
    """Vertex -> face incidence of a triangle list: (rowptr [V+1], list [3F]) int32, faces ascending."""
    fv = faces
    dev = fv.device
    F, V = fv.shape[0], int(num_vertices)
    rowptr = torch.empty(V + 1, dtype=torch.int32, device=dev)
    lst = torch.empty(max(3 * F, 1), dtype=torch.int32, device=dev)
    ws = L.workspace(L.size_query('geobi_vertex_faces_ws_bytes', F, V), dev)
    L.call('geobi_vertex_faces', L.ptr(fv), F, V, L.ptr(rowptr), L.ptr(lst), L.ptr(ws), ws.numel(), L.stream())
    return rowptr, lst[:3 * F]


def vf_padded32(rowptr, lst, num_vertices):
    """openmesh ``vf_indices`` as int32: [V, max_valence], -1 padded (what the device ring growth walks)."""
    dev = rowptr.device
    m = torch.zeros(1, dtype=torch.int32, device=dev)
    L.call('geobi_max_degree', L.ptr(rowptr), int(num_vertices), L.ptr(m), L.stream())
    maxval = max(L.read_i32(m, 1)[0], 1)
    vf = torch.empty((int(num_vertices), maxval), dtype=torch.int32, device=dev)
    L.call('geobi_vf_padded', L.ptr(rowptr), L.ptr(lst), int(num_vertices), maxval, L.ptr(vf), L.stream())
    return vf


def vf_padded(rowptr, lst, num_vertices):
    """openmesh ``vf_indices``: [V, max_valence] int64, -1 padded (what update_position2 takes)."""
    return vf_padded32(rowptr, lst, num_vertices).long()


def mesh_normals(points, faces, rowptr, lst):
    """-> (face normals [F,3], face centroids [F,3], vertex normals [V,3]) fp32."""
    dev = points.device
    F, V = faces.shape[0], points.shape[0]
    fn = torch.empty((F, 3), dtype=torch.float32, device=dev)
    cen = torch.empty((F, 3), dtype=torch.float32, device=dev)
    vn = torch.empty((V, 3), dtype=torch.float32, device=dev)
    L.call('geobi_mesh_normals', L.ptr(points), L.ptr(faces), F, V, L.ptr(rowptr), L.ptr(lst), L.ptr(fn),
           L.ptr(cen), L.ptr(vn), L.stream())
    return fn, cen, vn


def ring_graph(kind, faces, rowptr, lst, num_nodes):
    """kind 0: vertex graph, 1: facet graph -> loop-free symmetric ``Graph`` ((row, col)-sorted CSR)."""
    dev = faces.device
    n = int(num_nodes)
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ws = L.workspace(L.size_query('geobi_ring_graph_ws_bytes', n), dev)
    L.call('geobi_ring_graph_count', kind, L.ptr(faces), L.ptr(rowptr), L.ptr(lst), n, L.ptr(rp), L.ptr(ws),
           ws.numel(), L.stream())
    E = L.read_i32(rp[n:n + 1], 1)[0]                       # one host read per graph (sizes the column array)
    col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
    if E > 0:         # a graph without edges (the facet graph of a one-face patch: found by tools/fuzz_mesh.py) has nothing to fill
        L.call('geobi_ring_graph_fill', kind, L.ptr(faces), L.ptr(rowptr), L.ptr(lst), n, L.ptr(rp), L.ptr(col),
               L.stream())
    g = Graph(n, dev)
    g.rowptr_out, g.col_out, g.E = rp, col, E
    g.symmetric = True                                      # sharing a face / a vertex is a symmetric relation
    return g


def calc_weight(pos, normal, graph, extra_zero_edges=None, want_mean=False):
    """data_util.calc_weight over the loop-free edges of ``graph`` (CSR order).  The reference's edge list
    also holds one zero-length self loop per node, which enters its mean edge length: ``extra_zero_edges``
    (default N) adds them to the denominator."""
    dev = pos.device
    extra = graph.N if extra_zero_edges is None else int(extra_zero_edges)
    w = torch.empty(max(graph.E, 1), dtype=torch.float32, device=dev)[:graph.E]
    mean = torch.zeros(1, dtype=torch.float32, device=dev) if want_mean else None
    ws = L.workspace(L.size_query('geobi_calc_weight_ws_bytes'), dev)
    L.call('geobi_calc_weight', L.ptr(pos), L.ptr(normal), L.ptr(graph.ensure_rows()), L.ptr(graph.col_out), graph.E,
           extra, L.ptr(w), L.ptr(mean), L.ptr(ws), ws.numel(), L.stream())
    return (w, mean) if want_mean else w


def mean_edge_length(pos, graph):
    """Mean length of the mesh edges (center_and_scale's 1/scale, data_util.py:201-230) -> device scalar."""
    dev = pos.device
    mean = torch.zeros(1, dtype=torch.float32, device=dev)
    ws = L.workspace(L.size_query('geobi_calc_weight_ws_bytes'), dev)
    L.call('geobi_calc_weight', L.ptr(pos), None, L.ptr(graph.ensure_rows()), L.ptr(graph.col_out), graph.E, 0, None,
           L.ptr(mean), L.ptr(ws), ws.numel(), L.stream())
    return mean


def _reference_coo(graph, weight, normal, loops_inline):
    """The reference's COO + weights for a loop-free sorted graph: self loops appended (vertex graph,
    add_self_loops) or merged in row-major order (facet graph, coalesce).  calc_weight gives a loop
    clamp(n.n, 1e-3) * exp(0): 1 for unit normals, 1e-3 for the zero normal of a face-less vertex."""
    dev = graph.device
    n = graph.N
    row, col = graph.ensure_rows().long(), graph.col_out.long()
    loops = torch.arange(n, device=dev)
    r, c = torch.cat([row, loops]), torch.cat([col, loops])
    w = torch.cat([weight, (normal * normal).sum(1).clamp(min=0.001)])
    if loops_inline:
        order = torch.argsort(r * n + c)
        r, c, w = r[order], c[order], w[order]
    return torch.stack([r, c], 0), w


def calc_weight_parts(pos, normal, graph, node_ptr):
    """calc_weight over a disjoint union of meshes: ``node_ptr`` (device int32 [P+1]) cuts the nodes into parts and every
    part is normalised by ITS OWN mean edge length (geobi_calc_weight_parts: bit-identical to the part alone)."""
    dev = pos.device
    n_parts = int(node_ptr.numel()) - 1
    w = torch.empty(max(graph.E, 1), dtype=torch.float32, device=dev)[:graph.E]
    ws = L.workspace(L.size_query('geobi_calc_weight_parts_ws_bytes', n_parts), dev)
    L.call('geobi_calc_weight_parts', L.ptr(pos), L.ptr(normal), L.ptr(graph.rowptr_out), L.ptr(graph.ensure_rows()),
           L.ptr(graph.col_out), graph.E, L.ptr(node_ptr), n_parts, L.ptr(w), L.ptr(ws), ws.numel(), L.stream())
    return w


def build_dual_data(points_noisy, faces, points_gt=None, name='mesh', data_type='Synthetic', device=None,
                    reference_layout=False, centroid=None, scale=None, trusted_faces=False, want_vf=True, parts=None):
    """(points [V,3], faces [F,3]) -> (data_v, data_f) as process_one_submesh + post_processing emit them,
    computed on the device.  Same fields as ``meshgen.build_dual_data`` (incl. ``data_v.meta``).
    ``centroid`` [1,3] / ``scale``: normalisation of the WHOLE mesh when this is one patch of it
    (dataset.py:177-178 overwrite the patch's own values).
    trusted_faces: the face table was produced by this library (a patch of geobi_submesh): no range check.
    want_vf=False: no padded vf table in ``data_v.meta`` (a patch is never vertex-updated on its own).
    parts: (vertex_ptr, face_ptr) host int lists when the input is a DISJOINT UNION of meshes (the patches of one network
    pass, vertices / faces of part k in [ptr[k], ptr[k+1])): one preprocessing for all of them -- every step is local to a
    connected component except the bilateral weights' mean edge length, which is taken per part; ``mesh_ptr`` is set."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if not torch.cuda.is_available():
        raise L.GeobiError('meshprep.build_dual_data runs on the MI355X only (no CPU fallback); '
                           'meshgen.build_dual_data is the host-side generator')
    pts = torch.as_tensor(np.asarray(points_noisy) if not torch.is_tensor(points_noisy) else points_noisy)
    pts = pts.to(device=dev, dtype=torch.float32).contiguous()
    fv = _dev_i32(faces, dev)
    V, F = pts.shape[0], fv.shape[0]

    # A face table from outside is range-checked BEFORE any kernel walks it (a bad vertex id is a faulting gather); one
    # produced by this library (a patch of geobi_submesh) is not.
    if not trusted_faces and F > 0:
        lo, hi = L.read_i32(torch.cat([t.reshape(1) for t in torch.aminmax(fv)]))
        if lo < 0 or hi >= V:
            raise L.GeobiError('faces index vertices outside [0, %d)' % V)
    # Everything else whose size the host must know before it can allocate comes back in ONE read: both graphs' edge
    # counts and the largest valence (width of the padded vf table).  Until round 4 these were separate reads.
    rowptr_vf, lst = vertex_faces(fv, V)
    rp_v = torch.empty(V + 1, dtype=torch.int32, device=dev)
    rp_f = torch.empty(F + 1, dtype=torch.int32, device=dev)
    for kind, n, rp in ((0, V, rp_v), (1, F, rp_f)):
        ws = L.workspace(L.size_query('geobi_ring_graph_ws_bytes', n), dev)
        L.call('geobi_ring_graph_count', kind, L.ptr(fv), L.ptr(rowptr_vf), L.ptr(lst), n, L.ptr(rp), L.ptr(ws), ws.numel(),
               L.stream())
    size_words = [rp_v[V:V + 1], rp_f[F:F + 1]]
    if want_vf:
        m = torch.zeros(1, dtype=torch.int32, device=dev)
        L.call('geobi_max_degree', L.ptr(rowptr_vf), V, L.ptr(m), L.stream())
        size_words.append(m)
    sizes = L.read_i32(torch.cat(size_words))
    E_v, E_f = sizes[0], sizes[1]

    def finish_graph(kind, n, rp, E):
        col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        if E > 0:     # (a one-face patch has a facet graph without edges)
            L.call('geobi_ring_graph_fill', kind, L.ptr(fv), L.ptr(rowptr_vf), L.ptr(lst), n, L.ptr(rp), L.ptr(col), L.stream())
        g = Graph(n, dev)
        g.rowptr_out, g.col_out, g.E = rp, col, E
        g.symmetric = True                                  # sharing a face / a vertex is a symmetric relation
        return g
    fn, pos_f, vn = mesh_normals(pts, fv, rowptr_vf, lst)
    g_v = finish_graph(0, V, rp_v, E_v)
    g_f = finish_graph(1, F, rp_f, E_f)

    # center_and_scale, s_type 0: centroid = mean vertex, scale = 1 / mean mesh-edge length
    if centroid is None:
        cen = pts.mean(0, keepdim=True)
    else:
        cen = torch.as_tensor(centroid, dtype=torch.float32).reshape(1, 3).to(dev)
    sc = float((1.0 / mean_edge_length(pts, g_v)).item()) if scale is None else float(scale)

    if parts is None:
        ew_v = calc_weight(pts, vn, g_v)
        ew_f = calc_weight(pos_f, fn, g_f)
    else:
        ptr_dev = torch.tensor(list(parts[0]) + list(parts[1]), dtype=torch.int32).to(dev, non_blocking=True)
        nvp = len(parts[0])
        ew_v = calc_weight_parts(pts, vn, g_v, ptr_dev[:nvp])
        ew_f = calc_weight_parts(pos_f, fn, g_f, ptr_dev[nvp:])

    data_v = Data(torch.cat(((pts - cen) * sc, vn), 1), None, name=name + '-v')
    data_f = Data(torch.cat(((pos_f - cen) * sc, fn), 1), None, fv_indices=fv.long(), name=name + '-f')
    from .network import mark_face_table
    mark_face_table(data_f.fv_indices, fv, V)          # ids were range-checked above
    if reference_layout:
        ei_v, w_v = _reference_coo(g_v, ew_v, vn, loops_inline=False)
        ei_f, w_f = _reference_coo(g_f, ew_f, fn, loops_inline=True)
        data_v.edge_index, data_v.edge_weight = ei_v, w_v
        data_f.edge_index, data_f.edge_weight = ei_f, w_f
    else:
        data_v.set_graph(g_v); data_v.edge_weight = ew_v
        data_f.set_graph(g_f); data_f.edge_weight = ew_f
    data_v.depth_direction = None
    if data_type in ('Kinect_v1', 'Kinect_v2'):
        data_v.depth_direction = torch.nn.functional.normalize(pts, dim=1)
    if points_gt is not None:
        pg = torch.as_tensor(np.asarray(points_gt) if not torch.is_tensor(points_gt) else points_gt)
        pg = pg.to(device=dev, dtype=torch.float32).contiguous()
        data_v.y = (pg - cen) * sc
        gt_fn = torch.empty((F, 3), dtype=torch.float32, device=dev)
        gt_c = torch.empty((F, 3), dtype=torch.float32, device=dev)
        L.call('geobi_mesh_normals', L.ptr(pg), L.ptr(fv), F, V, None, None, L.ptr(gt_fn), L.ptr(gt_c), None, L.stream())
        data_f.y = gt_fn
    vf = None
    if want_vf:
        maxval = max(sizes[2], 1)
        vf = torch.empty((V, maxval), dtype=torch.int32, device=dev)
        L.call('geobi_vf_padded', L.ptr(rowptr_vf), L.ptr(lst), V, maxval, L.ptr(vf), L.stream())
        vf = vf.long()
    data_v.meta = {'centroid': cen, 'scale': sc, 'vf_indices': vf, 'incidence': (rowptr_vf, lst)}
    if parts is not None:
        data_v.mesh_ptr = torch.tensor(list(parts[0]), dtype=torch.long)
        data_f.mesh_ptr = torch.tensor(list(parts[1]), dtype=torch.long)
    return data_v, data_f
