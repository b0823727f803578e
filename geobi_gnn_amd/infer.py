"""Inference counterpart of /root/reference/code/test_dual.py:18-87 (predict_one_submesh / predict_one)
without the OBJ file IO: network forward under no_grad, de-normalisation, 60-sweep vertex update on the
device, and the two angular errors the reference prints."""
import torch

from . import network
from .data_util import computer_face_normal, update_position2


def predict_one_submesh(net, dual_data):
    """test_dual.py:18-22."""
    with torch.no_grad():
        vert_p, norm_p, _ = net((dual_data[0].shallow_copy(), dual_data[1].shallow_copy()))
    return vert_p, norm_p


def predict_one(net, data_v, data_f, centroid, scale, vf_indices, n_iter=60, gt_normals=None):
    """test_dual.py:44-87 for a mesh that fits one patch.

    Returns dict(Vp, Np, V_updated, angle1, angle2): Vp de-normalised predicted vertices, Np predicted
    unit normals, V_updated after update_position2; angle1 = error_n(Np, GT), angle2 = error_n(normals of
    the updated mesh, GT) when ground-truth normals are given (degrees)."""
    Vp, Np = predict_one_submesh(net, (data_v, data_f))
    Vp = Vp / scale + centroid.to(Vp.device)
    dd = getattr(data_v, 'depth_direction', None) if net.force_depth else None
    Vu = update_position2(Vp, data_f.fv_indices, vf_indices.to(Vp.device), Np, n_iter=n_iter, depth_direction=dd)
    out = {'Vp': Vp, 'Np': Np, 'V_updated': Vu, 'angle1': None, 'angle2': None}
    if gt_normals is not None:
        out['angle1'] = float(network.error_n(Np, gt_normals))
        out['angle2'] = float(network.error_n(computer_face_normal(Vu, data_f.fv_indices), gt_normals))
    return out
