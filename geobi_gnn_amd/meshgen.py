"""Synthetic meshes and the two input graphs of the hot path, without openmesh/PyG.

The real datasets are external downloads (/root/reference/README.md:7), so benchmarks
and parity tests run on class-I geodesic icospheres (F = 20 n^2, V = 10 n^2 + 2) with
Gaussian noise along the vertex normals.  The graph assembly
below restates, for a plain (points, faces) pair:

* /root/reference/code/dataset.py:197-243  process_one_submesh (vertex graph =
  undirected mesh edges + appended self loops; facet graph = faces sharing a vertex,
  self loops inline, row-major sorted),
* /root/reference/code/data_util.py:383-398 calc_weight (bilateral edge weight, with
  its squared-length over un-squared-mean quirk),
* /root/reference/code/data_util.py:201-230 center_and_scale (s_type 0),
* /root/reference/code/dataset.py:246-269  post_processing (feature assembly).

Everything here is host-side numpy/torch preprocessing (cached by the reference as
``.pt`` files); it is not part of the accelerated path but defines its input contract.
"""
import numpy as np
import torch

from .data import Data


# --------------------------------------------------------------------------- meshes
def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
                  [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                  [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                  [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    return v, f


_ICOSPHERES = {}


def icosphere(n):
    """Class-I geodesic sphere of frequency ``n``: returns (points [V,3] f64, faces [F,3] i64)."""
    if n not in _ICOSPHERES:
        pts, faces = _icosphere(n)
        pts.setflags(write=False)
        faces.setflags(write=False)
        _ICOSPHERES[n] = (pts, faces)           # the lattice walk below is a Python loop over 20 n^2 faces
    pts, faces = _ICOSPHERES[n]
    return pts.copy(), faces.copy()


def _icosphere(n):
    assert n >= 1
    v0, f0 = _icosahedron()
    # barycentric lattice on every base triangle; vertices are identified through an
    # exact integer key (sorted (base-vertex, weight) pairs), so shared edges/corners merge.
    keys = {}
    pts = []
    faces = []

    def vid(tri, i, j, k):
        w = [(int(tri[0]), i), (int(tri[1]), j), (int(tri[2]), k)]
        key = tuple(sorted((a, b) for a, b in w if b > 0))
        idx = keys.get(key)
        if idx is None:
            idx = len(pts)
            keys[key] = idx
            p = (i * v0[tri[0]] + j * v0[tri[1]] + k * v0[tri[2]]) / float(n)
            pts.append(p / np.linalg.norm(p))
        return idx

    for tri in f0:
        for i in range(n):
            for j in range(n - i):
                k = n - i - j
                # "up" triangle
                a = vid(tri, k, i, j)
                b = vid(tri, k - 1, i + 1, j)
                c = vid(tri, k - 1, i, j + 1)
                faces.append((a, b, c))
                if j + i < n - 1:  # "down" triangle
                    d = vid(tri, k - 2, i + 1, j + 1)
                    faces.append((b, d, c))
    pts = np.asarray(pts, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    # consistent outward orientation
    fn = np.cross(pts[faces[:, 1]] - pts[faces[:, 0]], pts[faces[:, 2]] - pts[faces[:, 0]])
    flip = (fn * pts[faces].mean(1)).sum(1) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]
    assert faces.shape[0] == 20 * n * n and pts.shape[0] == 10 * n * n + 2
    return pts, faces


def face_normals(points, faces):
    fn = np.cross(points[faces[:, 1]] - points[faces[:, 0]], points[faces[:, 2]] - points[faces[:, 0]])
    d = np.clip(np.linalg.norm(fn, axis=1, keepdims=True), 1e-12, None)
    return fn / d


def vertex_normals(points, faces, fn=None):
    """Normalised sum of incident face normals (openmesh ``update_vertex_normals`` default)."""
    if fn is None:
        fn = face_normals(points, faces)
    vn = np.zeros_like(points)
    for c in range(3):
        np.add.at(vn, faces[:, c], fn)
    d = np.clip(np.linalg.norm(vn, axis=1, keepdims=True), 1e-12, None)
    return vn / d


def mesh_edges(faces):
    """Unique undirected edges [M,2] (the role of openmesh ``ev_indices``)."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    e = np.sort(e, axis=1)
    return np.unique(e, axis=0)


def vertex_faces(faces, num_vertices):
    """Padded vertex->face table [V, max_valence], -1 filled (openmesh ``vf_indices``)."""
    F = faces.shape[0]
    vert = faces.reshape(-1)
    face = np.repeat(np.arange(F), 3)
    order = np.argsort(vert, kind='stable')
    vert, face = vert[order], face[order]
    counts = np.bincount(vert, minlength=num_vertices)
    start = np.concatenate([[0], np.cumsum(counts)[:-1]])
    slot = np.arange(vert.shape[0]) - start[vert]
    vf = -np.ones((num_vertices, int(counts.max())), dtype=np.int64)
    vf[vert, slot] = face
    return vf


def noisy_icosphere(n, sigma=0.2, seed=0):
    """Clean sphere + copy jittered along vertex normals by sigma * mean-edge-length * N(0,1)."""
    pts, faces = icosphere(n)
    ev = mesh_edges(faces)
    mean_len = np.linalg.norm(pts[ev[:, 0]] - pts[ev[:, 1]], axis=1).mean()
    rng = np.random.default_rng(seed)
    vn = vertex_normals(pts, faces)
    noisy = pts + vn * (sigma * mean_len * rng.standard_normal((pts.shape[0], 1)))
    return noisy.astype(np.float32), pts.astype(np.float32), faces


# --------------------------------------------------------------------------- graphs
def coalesce_index(row, col, num_nodes):
    """Sorted unique (row, col) pairs -- torch_sparse.coalesce without values."""
    key = np.unique(row.astype(np.int64) * num_nodes + col.astype(np.int64))
    return key // num_nodes, key % num_nodes


def vertex_graph_index(faces, num_vertices):
    """dataset.py:211-213: to_undirected(ev.T) then add_self_loops (loops appended last)."""
    ev = mesh_edges(faces)
    row = np.concatenate([ev[:, 0], ev[:, 1]])
    col = np.concatenate([ev[:, 1], ev[:, 0]])
    row, col = coalesce_index(row, col, num_vertices)
    loops = np.arange(num_vertices, dtype=np.int64)
    return np.stack([np.concatenate([row, loops]), np.concatenate([col, loops])], 0)


def facet_graph_index(faces, vf):
    """data_util.py:436-456 build_facet_graph: faces sharing >= 1 vertex, self loops inline."""
    F = faces.shape[0]
    nbr = vf[faces, :].reshape(F, -1)                 # [F, 3*maxval]
    row = np.repeat(np.arange(F), nbr.shape[1])
    col = nbr.reshape(-1)
    ok = col > -1
    row, col = coalesce_index(row[ok], col[ok], F)
    return np.stack([row, col], 0)


def bilateral_edge_weight(pos, normal, edge_index):
    """The static edge weight the dataset stores (what data_util.py:383-398 defines):
    max(n_s . n_t, 1e-3) * exp(-|p_s - p_t|^2 / (2 * mean edge length)), the mean running over every
    listed edge, loops included.  Host-side generator only; the device path is meshprep.calc_weight
    (csrc/meshprep.hip: calc_weight_kernel), pinned by tests/golden/pure_functions.npz."""
    src, dst = edge_index[0], edge_index[1]
    sq_len = (pos[src] - pos[dst]).pow(2).sum(1)
    two_sigma = 2 * sq_len.sqrt().mean() - 1e-12
    cos_angle = (normal[src] * normal[dst]).sum(1).clamp_min(1e-3)
    return cos_angle * torch.exp(-sq_len / two_sigma)


def build_dual_data(points_noisy, faces, points_gt=None, name='mesh', data_type='Synthetic'):
    """(points, faces) -> (data_v, data_f) exactly as process_one_submesh + post_processing emit them."""
    pn = np.asarray(points_noisy, dtype=np.float32)
    V = pn.shape[0]
    ev = mesh_edges(faces)
    # center_and_scale, s_type 0 (data_util.py:201-230)
    centroid = pn.mean(0, keepdims=True)
    pc = pn - centroid
    scale = np.float32(1.0) / (((pc[ev[:, 0]] - pc[ev[:, 1]]) ** 2).sum(1) ** 0.5).mean()

    fn = face_normals(pn.astype(np.float64), faces).astype(np.float32)
    vn = vertex_normals(pn.astype(np.float64), faces).astype(np.float32)
    vf = vertex_faces(faces, V)

    fv_t = torch.from_numpy(faces).long()
    pos_v = torch.from_numpy(pn).float()
    normal_v = torch.from_numpy(vn).float()
    ei_v = torch.from_numpy(vertex_graph_index(faces, V)).long()
    ew_v = bilateral_edge_weight(pos_v, normal_v, ei_v).float()

    pos_f = pos_v[fv_t].mean(1).float()
    normal_f = torch.from_numpy(fn).float()
    ei_f = torch.from_numpy(facet_graph_index(faces, vf)).long()
    ew_f = bilateral_edge_weight(pos_f, normal_f, ei_f).float()

    cen = torch.from_numpy(centroid).float()
    sc = float(scale)
    data_v = Data(torch.cat(((pos_v - cen) * sc, normal_v), 1), ei_v, edge_weight=ew_v, name=name + '-v')
    data_f = Data(torch.cat(((pos_f - cen) * sc, normal_f), 1), ei_f, edge_weight=ew_f,
                  fv_indices=fv_t, name=name + '-f')
    data_v.depth_direction = None
    if data_type in ('Kinect_v1', 'Kinect_v2'):
        data_v.depth_direction = torch.nn.functional.normalize(pos_v, dim=1).float()
    if points_gt is not None:
        pg = np.asarray(points_gt, dtype=np.float32)
        data_v.y = (torch.from_numpy(pg).float() - cen) * sc
        data_f.y = torch.from_numpy(face_normals(pg.astype(np.float64), faces).astype(np.float32)).float()
    # kept for the post-network vertex update / de-normalisation (test_dual.py:63-72)
    data_v.meta = {'centroid': cen, 'scale': sc, 'vf_indices': torch.from_numpy(vf).long()}
    return data_v, data_f


def synthetic_dual_data(n, sigma=0.2, seed=0, data_type='Synthetic'):
    noisy, clean, faces = noisy_icosphere(n, sigma, seed)
    return build_dual_data(noisy, faces, clean, name='ico%d_s%d' % (n, seed), data_type=data_type)
