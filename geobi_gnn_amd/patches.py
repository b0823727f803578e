"""Patch split / merge for meshes larger than one network pass.

* split  /root/reference/code/dataset.py:156-193 (``process_one_data``: seed = face farthest from the
         centroid, grow ``submesh_size`` faces ring by ring, next seed = farthest unvisited face) with
         /root/reference/code/data_util.py:55-84 ``mesh_get_neighbor_np`` and :318-336 ``get_submesh``
* merge  /root/reference/code/test_dual.py:49-61 (sum overlapping predictions, divide by the visit count,
         re-normalise the normals) followed by the 60-sweep vertex update (:63-72)

The ordered ring growth runs on the host (C++, ``geobi_patch_grow_host``); vertex renumbering, graph
construction, the network and the merge run on the device.  The reference walks openmesh's ``vf_indices``
rows; here the incidence lists are in ascending face order, so which faces the LAST, partial ring of a
patch contributes can differ from an openmesh run (whole rings do not depend on the order).
"""
import ctypes

import numpy as np
import torch

from . import _lib as L
from . import meshprep, network
from .data_util import computer_face_normal, update_position2
from .infer import predict_one_submesh


def patch_grow(fv_host, rowptr_host, list_host, seed, neighbor_count=None, ring_count=None):
    """data_util.mesh_get_neighbor_np on host arrays (int32, C-contiguous) -> face ids in visiting order."""
    F = fv_host.shape[0]
    out = np.empty(F, dtype=np.int32)
    n = ctypes.c_int64(0)
    L.call('geobi_patch_grow_host', fv_host.ctypes.data, rowptr_host.ctypes.data, list_host.ctypes.data, F, int(seed),
           int(neighbor_count or 0), int(ring_count or 0), out.ctypes.data, ctypes.byref(n))
    return out[:n.value]


def submesh(fv, sel, num_vertices):
    """data_util.get_submesh on the device: sel [n] int32 face ids -> (V_idx [nv] int32, F_sub [n,3] int32)."""
    dev = fv.device
    n, V = int(sel.shape[0]), int(num_vertices)
    v_idx = torch.empty(min(V, 3 * n), dtype=torch.int32, device=dev)
    f_sub = torch.empty((n, 3), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = L.workspace(L.size_query('geobi_submesh_ws_bytes', n, V), dev)
    L.call('geobi_submesh', L.ptr(fv), L.ptr(sel), n, V, L.ptr(v_idx), L.ptr(f_sub), L.ptr(count), L.ptr(ws),
           ws.numel(), L.stream())
    return v_idx[:L.read_i32(count, 1)[0]], f_sub


def split_patches(points, fv, submesh_size, incidence=None, keep=None):
    """Generator over the patches of dataset.py:156-193: yields (select_faces, V_idx, F_sub) device int32
    tensors.  points [V,3] fp32 and fv [F,3] int32 live on the device.
    keep: optional callable(k) -> bool.  Patch k is still grown (the next seed depends on every earlier patch) but
    when keep(k) is false nothing is built on the device and None is yielded -- how the ranks of a multi-GPU
    inference each take their share of one mesh's patches."""
    V, F = points.shape[0], fv.shape[0]
    rowptr, lst = incidence if incidence is not None else meshprep.vertex_faces(fv, V)
    fv_h = fv.cpu().numpy()
    rp_h, ls_h = rowptr.cpu().numpy(), lst.cpu().numpy()
    centroid = points.mean(0, keepdim=True)
    face_cent = points[fv.long()].mean(1)
    d2 = ((face_cent - centroid) ** 2).sum(1).cpu().numpy()
    flag = np.zeros(F, dtype=bool)
    seed = int(np.argmax(d2))
    k = 0
    while True:
        sel_h = patch_grow(fv_h, rp_h, ls_h, seed, neighbor_count=submesh_size)
        flag[sel_h] = True
        if keep is None or keep(k):
            sel = torch.from_numpy(sel_h).to(fv.device)
            v_idx, f_sub = submesh(fv, sel, V)
            yield sel, v_idx, f_sub
        else:
            yield None
        k += 1
        left = np.where(~flag)[0]
        if left.size == 0:
            break
        seed = int(left[np.argmax(d2[left])])


def _union_dual(duals):
    """Several device-built (data_v, data_f) pairs -> one pair over the disjoint union of their graphs (CSR
    concatenated directly); returns the pair and the per-patch (vertex, face) row ranges."""
    from .data import union_batch_graphs
    data_v, data_f = union_batch_graphs(duals)
    pv, pf = data_v.mesh_ptr.tolist(), data_f.mesh_ptr.tolist()
    return (data_v, data_f), list(zip(pv[:-1], pv[1:])), list(zip(pf[:-1], pf[1:]))


def predict_mesh(net, points, faces, sub_size=20000, n_iter=60, data_type='Synthetic', gt_points=None, patch_batch=8,
                 distributed=None):
    """test_dual.py:24-87 without the OBJ IO, for a mesh of any size: preprocessing, patch split when
    F > sub_size, network, merge, de-normalisation, vertex update -- all device-resident.  The reference runs
    the patches one by one; here `patch_batch` of them go through the network as one disjoint-union graph
    (independent components: same results, fewer and better-filled launches).

    Multi-GPU: with torch.distributed initialised (``distributed=None`` picks that up; False turns it
    off) every rank calls this with the same mesh; the patches are dealt round-robin over the ranks
    (parallel.owns_patch), each rank merges its own into its Vp / Np / visit-count sums, ONE reduction to rank 0
    (parallel.reduce_patch_sums) adds them, and rank 0 finalises and runs the vertex update.  No graph is split
    across GPUs and nothing but those three sums crosses xGMI.  Other ranks get Vp = Np = V_updated = None.

    Returns dict(Vp, Np, V_updated, n_patches, angle1, angle2)."""
    from . import parallel
    rank, world = parallel.rank_world() if distributed is None or distributed else (0, 1)
    dev = next(net.parameters()).device
    pts = torch.as_tensor(np.asarray(points) if not torch.is_tensor(points) else points)
    pts = pts.to(device=dev, dtype=torch.float32).contiguous()
    fv = torch.as_tensor(np.asarray(faces) if not torch.is_tensor(faces) else faces).to(device=dev, dtype=torch.int32)
    fv = fv.contiguous()
    V, F = pts.shape[0], fv.shape[0]
    rowptr, lst = meshprep.vertex_faces(fv, V)
    g_v = meshprep.ring_graph(0, fv, rowptr, lst, V)
    centroid = pts.mean(0, keepdim=True)
    scale = float((1.0 / meshprep.mean_edge_length(pts, g_v)).item())

    if F <= sub_size:
        n_patches = 1
        if rank != 0:                  # one patch: nothing to share out
            return {'Vp': None, 'Np': None, 'V_updated': None, 'n_patches': 1, 'angle1': None, 'angle2': None}
        dual = meshprep.build_dual_data(pts, fv, name='mesh', data_type=data_type, device=dev)
        Vp, Np = predict_one_submesh(net, dual)
        Vp = Vp / scale + centroid
    else:
        Vp = torch.zeros((V, 3), dtype=torch.float32, device=dev)
        Np = torch.zeros((F, 3), dtype=torch.float32, device=dev)
        sum_v = torch.zeros(V, dtype=torch.int32, device=dev)
        n_patches = 0
        pending = []

        def flush():
            if not pending:
                return
            if len(pending) == 1:
                dual, vr, fr = pending[0][2], [(0, pending[0][1].shape[0])], [(0, pending[0][0].shape[0])]
            else:
                dual, vr, fr = _union_dual([p[2] for p in pending])
            vert_p, norm_p = predict_one_submesh(net, dual)
            for (sel, v_idx, _), (v0, v1), (f0, f1) in zip(pending, vr, fr):
                L.call('geobi_patch_accumulate', L.ptr(vert_p[v0:v1]), L.ptr(norm_p[f0:f1]), L.ptr(v_idx), L.ptr(sel),
                       v_idx.shape[0], sel.shape[0], L.ptr(Vp), L.ptr(Np), L.ptr(sum_v), L.stream())
            del pending[:]

        keep = None if world == 1 else (lambda k: parallel.owns_patch(k, rank, world))
        for part in split_patches(pts, fv, sub_size, incidence=(rowptr, lst), keep=keep):
            n_patches += 1
            if part is None:
                continue
            sel, v_idx, f_sub = part
            dual = meshprep.build_dual_data(pts[v_idx.long()], f_sub, name='patch%d' % (n_patches - 1),
                                            data_type=data_type, device=dev, centroid=centroid, scale=scale)
            pending.append((sel, v_idx, dual))
            if len(pending) >= max(1, int(patch_batch)):
                flush()
        flush()
        if world > 1:
            parallel.reduce_patch_sums([Vp, Np, sum_v], dst=0)
            if rank != 0:
                return {'Vp': None, 'Np': None, 'V_updated': None, 'n_patches': n_patches, 'angle1': None,
                        'angle2': None}
        c = centroid.reshape(-1).tolist()
        L.call('geobi_patch_finalize', L.ptr(Vp), L.ptr(Np), L.ptr(sum_v), V, F, scale, c[0], c[1], c[2], L.stream())

    dd = torch.nn.functional.normalize(pts, dim=1) if data_type in ('Kinect_v1', 'Kinect_v2') else None
    vf = meshprep.vf_padded(rowptr, lst, V)
    Vu = update_position2(Vp, fv, vf, Np, n_iter=n_iter, depth_direction=dd)
    out = {'Vp': Vp, 'Np': Np, 'V_updated': Vu, 'n_patches': n_patches, 'angle1': None, 'angle2': None}
    if gt_points is not None:
        gt = torch.as_tensor(np.asarray(gt_points) if not torch.is_tensor(gt_points) else gt_points)
        gt = gt.to(device=dev, dtype=torch.float32).contiguous()
        Nt = computer_face_normal(gt, fv)
        out['angle1'] = float(network.error_n(Np, Nt))
        out['angle2'] = float(network.error_n(computer_face_normal(Vu, fv), Nt))
    return out
