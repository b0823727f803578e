"""Patch split / merge on the device against the fixture produced by the
reference's data_util.mesh_get_neighbor_np / get_submesh (tests/golden/patches_n8.npz)."""
import numpy as np
import pytest
import torch

from helpers import load_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    return torch.device('cuda:0')


def test_split_matches_reference_fixture(dev):
    from geobi_gnn_amd import patches
    fx = load_fixture('patches_n8.npz')
    pts = torch.from_numpy(fx['points']).to(dev)
    fv = torch.from_numpy(fx['faces']).to(dev).int().contiguous()
    got = list(patches.split_patches(pts, fv, int(fx['sub_size'])))
    assert len(got) == len(fx['seeds'])
    fo = vo = 0
    for (sel, v_idx, f_sub), (nf, nv) in zip(got, fx['sizes']):
        assert np.array_equal(sel.cpu().numpy(), fx['select_faces'][fo:fo + nf])
        assert np.array_equal(v_idx.cpu().numpy(), fx['v_idx'][vo:vo + nv])
        assert np.array_equal(f_sub.cpu().numpy(), fx['f_sub'][fo:fo + nf])
        # the renumbered faces index the gathered vertices back to the original triangles
        assert np.array_equal(v_idx.cpu().numpy()[f_sub.cpu().numpy()], fx['faces'][sel.cpu().numpy()])
        fo += nf
        vo += nv


def test_predict_mesh_merges_like_the_reference(dev):
    """test_dual.py:49-61 restated with torch indexing beside the device merge."""
    from geobi_gnn_amd import meshprep, network, patches
    from geobi_gnn_amd.infer import predict_one_submesh
    fx = load_fixture('patches_n8.npz')
    pts = torch.from_numpy(fx['points']).to(dev)
    fv = torch.from_numpy(fx['faces']).to(dev).int().contiguous()
    V, F = pts.shape[0], fv.shape[0]
    torch.manual_seed(0)
    net = network.DualGNN().to(dev).eval()
    sub = int(fx['sub_size'])
    out = patches.predict_mesh(net, pts, fv, sub_size=sub, n_iter=5, gt_points=fx['clean'])
    assert out['n_patches'] == len(fx['seeds'])
    # patches through the network one by one or several per pass (disjoint union): the same numbers
    for pb in (1, 3):
        alt = patches.predict_mesh(net, pts, fv, sub_size=sub, n_iter=5, patch_batch=pb)
        assert torch.equal(alt['Np'], out['Np']) and torch.equal(alt['Vp'], out['Vp'])

    rowptr, lst = meshprep.vertex_faces(fv, V)
    g_v = meshprep.ring_graph(0, fv, rowptr, lst, V)
    centroid = pts.mean(0, keepdim=True)
    scale = float((1.0 / meshprep.mean_edge_length(pts, g_v)).item())
    sum_v = torch.zeros((V, 1), device=dev)
    Vp = torch.zeros((V, 3), device=dev)
    Np = torch.zeros((F, 3), device=dev)
    fo = vo = 0
    for nf, nv in fx['sizes']:
        sel = torch.from_numpy(fx['select_faces'][fo:fo + nf]).to(dev).long()
        v_idx = torch.from_numpy(fx['v_idx'][vo:vo + nv]).to(dev).long()
        f_sub = torch.from_numpy(fx['f_sub'][fo:fo + nf]).to(dev).int()
        dual = meshprep.build_dual_data(pts[v_idx], f_sub, device=dev, centroid=centroid, scale=scale)
        vert_p, norm_p = predict_one_submesh(net, dual)
        sum_v[v_idx] += 1
        Vp[v_idx] += vert_p
        Np[sel] += norm_p
        fo += nf
        vo += nv
    Vp = Vp / sum_v / scale + centroid
    Np = torch.nn.functional.normalize(Np, dim=1)
    assert float(sum_v.min()) >= 1 and float(sum_v.max()) > 1          # patches overlap
    assert float((out['Vp'] - Vp).abs().max()) <= 1e-5 * float(Vp.abs().max())
    assert float((out['Np'] - Np).abs().max()) <= 1e-5
    assert torch.isfinite(out['V_updated']).all()
    assert out['angle1'] is not None and 0.0 <= out['angle1'] <= 180.0

    # a mesh that fits one pass takes the single-patch branch
    one = patches.predict_mesh(net, pts, fv, sub_size=F, n_iter=5)
    assert one['n_patches'] == 1 and one['Np'].shape == (F, 3)
