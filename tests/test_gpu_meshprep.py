"""Device-side mesh preprocessing against the host-side generator
(meshgen, numpy/torch CPU restatement of dataset.py:197-269) and against the golden vectors that the
reference's own data_util.calc_weight produced (tests/golden/pure_functions.npz)."""
import numpy as np
import pytest
import torch

from helpers import load_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    return torch.device('cuda:0')


def _strip_loops_sorted(ei, w=None):
    ei = np.asarray(ei)
    keep = ei[0] != ei[1]
    r, c = ei[0][keep], ei[1][keep]
    n = int(ei.max()) + 1
    order = np.argsort(r.astype(np.int64) * n + c, kind='stable')
    out = np.stack([r[order], c[order]], 0)
    return (out, None) if w is None else (out, np.asarray(w)[keep][order])


def _fan_mesh(k=40):
    """One vertex of valence k (> the 32 a wave-per-node scheme would hold) + an isolated vertex."""
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    ring = np.stack([np.cos(ang), np.sin(ang), 0.1 * np.sin(3 * ang)], 1)
    pts = np.concatenate([[[0, 0, 0.3]], ring, [[5.0, 5.0, 5.0]]], 0).astype(np.float32)
    faces = np.array([[0, 1 + i, 1 + (i + 1) % k] for i in range(k)], dtype=np.int64)
    return pts, faces


def _holey_icosphere(n=6, seed=3):
    from geobi_gnn_amd import meshgen
    noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed)
    rng = np.random.default_rng(seed)
    keep = rng.random(faces.shape[0]) > 0.15            # holes and boundaries, some vertices lose all faces
    return noisy, clean, faces[keep]


def _check_against_meshgen(dev, pts, faces, gt=None, tol_w=3e-6):
    from geobi_gnn_amd import meshgen, meshprep
    ref_v, ref_f = meshgen.build_dual_data(pts, faces, gt, name='m')
    dv, df = meshprep.build_dual_data(pts, faces, gt, name='m', device=dev, reference_layout=True)
    # exact reference layout: same int64 COO tensors
    assert torch.equal(dv.edge_index.cpu(), ref_v.edge_index)
    assert torch.equal(df.edge_index.cpu(), ref_f.edge_index)
    assert torch.equal(dv.meta['vf_indices'].cpu(), ref_v.meta['vf_indices'])
    assert torch.equal(df.fv_indices.cpu(), ref_f.fv_indices)
    np.testing.assert_allclose(dv.edge_weight.cpu().numpy(), ref_v.edge_weight.numpy(), rtol=tol_w, atol=1e-7)
    np.testing.assert_allclose(df.edge_weight.cpu().numpy(), ref_f.edge_weight.numpy(), rtol=tol_w, atol=1e-7)
    # features: [scaled position | normal]; positions are O(mesh size / edge length)
    scale = max(1.0, float(ref_v.x[:, :3].abs().max()))
    np.testing.assert_allclose(dv.x.cpu().numpy(), ref_v.x.numpy(), atol=2e-6 * scale)
    np.testing.assert_allclose(df.x.cpu().numpy(), ref_f.x.numpy(), atol=2e-6 * scale)
    assert abs(dv.meta['scale'] - ref_v.meta['scale']) <= 2e-6 * abs(ref_v.meta['scale'])
    if gt is not None:
        np.testing.assert_allclose(dv.y.cpu().numpy(), ref_v.y.numpy(), atol=2e-6 * scale)
        np.testing.assert_allclose(df.y.cpu().numpy(), ref_f.y.numpy(), atol=2e-6)
    # CSR layout (the default): loop-free sorted pairs, weights in that order
    cv, cf = meshprep.build_dual_data(pts, faces, gt, name='m', device=dev)
    for got, ref in ((cv, ref_v), (cf, ref_f)):
        ei, w = _strip_loops_sorted(ref.edge_index.numpy(), ref.edge_weight.numpy())
        assert np.array_equal(got.edge_index.cpu().numpy(), ei)
        np.testing.assert_allclose(got.edge_weight.cpu().numpy(), w, rtol=tol_w, atol=1e-7)
        g = got.graph()
        assert g.symmetric and g.E == ei.shape[1]
    return (dv, df), (cv, cf), (ref_v, ref_f)


@pytest.mark.parametrize('n', [3, 8, 16])
def test_icosphere_matches_host_generator(dev, n):
    from geobi_gnn_amd import meshgen
    noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=n)
    _check_against_meshgen(dev, noisy, faces, clean)


def test_irregular_meshes_match_host_generator(dev):
    pts, faces = _fan_mesh(40)
    _check_against_meshgen(dev, pts, faces)
    noisy, clean, faces = _holey_icosphere(6)
    _check_against_meshgen(dev, noisy, faces, clean)


def test_pinned_by_reference_calc_weight(dev):
    """pure_functions.npz: edge weights computed by the reference's data_util.calc_weight on this mesh
    (vertex graph incl. its V self loops); vf / normals from the openmesh stand-in."""
    from geobi_gnn_amd import meshprep
    fx = load_fixture('pure_functions.npz')
    pts = torch.from_numpy(fx['points']).to(dev)
    fv = torch.from_numpy(fx['faces']).to(dev).int().contiguous()
    V = pts.shape[0]
    rowptr, lst = meshprep.vertex_faces(fv, V)
    assert np.array_equal(meshprep.vf_padded(rowptr, lst, V).cpu().numpy(), fx['vf'].astype(np.int64))
    fn, cen, vn = meshprep.mesh_normals(pts, fv, rowptr, lst)
    np.testing.assert_allclose(fn.cpu().numpy(), fx['face_normal'], atol=2e-6)
    np.testing.assert_allclose(vn.cpu().numpy(), fx['vnormal'], atol=2e-6)
    g = meshprep.ring_graph(0, fv, rowptr, lst, V)
    ei, w = _strip_loops_sorted(fx['edge_index'], fx['calc_weight'])
    assert np.array_equal(torch.stack([g.ensure_rows(), g.col_out]).cpu().numpy(), ei)
    # the fixture's weights were computed from the fixture's vertex normals
    got = meshprep.calc_weight(pts, torch.from_numpy(fx['vnormal']).to(dev), g)
    np.testing.assert_allclose(got.cpu().numpy(), w, rtol=3e-6, atol=1e-7)
    loops = fx['calc_weight'][fx['edge_index'][0] == fx['edge_index'][1]]
    np.testing.assert_allclose(loops, 1.0, rtol=1e-6)          # what reference_layout=True appends
    # facet graph of the reference's own build_facet_graph (self loops inline) and its centre / scale
    dv, df = meshprep.build_dual_data(fx['points'], fx['faces'].astype(np.int64), device=dev, reference_layout=True)
    assert np.array_equal(df.edge_index.cpu().numpy(), fx['facet_graph_index'].astype(np.int64))
    np.testing.assert_allclose(dv.x[:, :3].cpu().numpy(), fx['centered_scaled'], rtol=0,
                               atol=2e-6 * np.abs(fx['centered_scaled']).max())
    assert abs(dv.meta['scale'] - float(fx['scale'])) <= 2e-6 * float(fx['scale'])


def test_network_on_device_built_inputs(dev):
    """The model gives the same answer on device-built inputs (CSR attached, no COO) as on the
    host-built tensors moved to the GPU."""
    from geobi_gnn_amd import meshgen, meshprep, network
    noisy, clean, faces = meshgen.noisy_icosphere(8, 0.2, seed=5)
    torch.manual_seed(0)
    net = network.DualGNN().to(dev).eval()
    hv, hf = meshgen.build_dual_data(noisy, faces, clean)
    hv, hf = hv.to(dev), hf.to(dev)
    cv, cf = meshprep.build_dual_data(noisy, faces, clean, device=dev)
    with torch.no_grad():
        v0, n0, _ = net((hv, hf))
        v1, n1, _ = net((cv, cf))
    assert float((v0 - v1).abs().max()) <= 2e-5 * max(1.0, float(v0.abs().max()))
    assert float((n0 - n1).abs().max()) <= 2e-5


def test_rejects_bad_input(dev):
    from geobi_gnn_amd import meshprep
    from geobi_gnn_amd._lib import GeobiError
    pts = np.zeros((4, 3), dtype=np.float32)
    with pytest.raises(GeobiError):
        meshprep.build_dual_data(pts, np.array([[0, 1, 7]]), device=dev)
