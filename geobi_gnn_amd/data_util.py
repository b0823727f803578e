"""Geometry helpers of the reference's ``data_util`` that sit directly on either side of the network.

* ``computer_face_normal``  /root/reference/code/data_util.py:182-198 (called inside DualGNN.forward)
* ``update_position2``      /root/reference/code/data_util.py:529-556 (called at test_dual.py:63-72 right
  after the network; the reference moves the prediction to the CPU for it -- here it stays on the GPU)
Same names and argument meaning; tensors must live on the MI355X.
"""
import torch

from . import _lib as L
from . import ops


def computer_face_normal(points, fv_indices):
    """points [N,3], fv_indices [M,3] -> unit face normals [M,3] (differentiable)."""
    L.require_device(points, 'points')
    fv32 = fv_indices.to(torch.int32).contiguous()
    cidx = ops.SegmentIndex(fv32.view(-1), points.shape[0]) if points.requires_grad else None
    xf = torch.zeros((fv32.shape[0], 6), dtype=torch.float32, device=points.device)
    if cidx is None:
        out = torch.empty((fv32.shape[0], 12), dtype=torch.float32, device=points.device)
        L.call('geobi_face_geom_fwd', L.ptr(points.contiguous()), L.ptr(fv32), L.ptr(xf), 6, fv32.shape[0],
               L.ptr(out), L.stream())
        return out[:, 9:12]
    return ops.FaceGeomFn.apply(points, xf, fv32, cidx)[:, 9:12]


def update_position2(points, fv_indices, vf_indices, face_normals, n_iter=20, depth_direction=None):
    """n_iter Jacobi sweeps moving every vertex onto the planes of its adjacent faces.

    points [N,3], fv_indices [F,3], vf_indices [N, max_valence] (-1 padded), face_normals [F,3]."""
    L.require_device(points, 'points')
    pts = points.detach().float().contiguous()
    fv32 = fv_indices.to(torch.int32).contiguous()
    vf32 = vf_indices.to(torch.int32).contiguous()
    nrm = face_normals.detach().float().contiguous()
    dd = None if depth_direction is None else depth_direction.detach().float().contiguous()
    V, F = pts.shape[0], fv32.shape[0]
    out = torch.empty_like(pts)
    ws = L.workspace(L.size_query('geobi_update_position_ws_bytes', V, F), pts.device)
    L.call('geobi_update_position2', L.ptr(pts), L.ptr(fv32), L.ptr(vf32), vf32.shape[1], L.ptr(nrm), L.ptr(dd), V, F,
           int(n_iter), L.ptr(out), L.ptr(ws), ws.numel(), L.stream())
    return out
