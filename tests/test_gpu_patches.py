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


def _host_incidence(faces, V):
    from geobi_gnn_amd import meshgen
    vf = meshgen.vertex_faces(faces.astype(np.int64), V)
    counts = (vf >= 0).sum(1)
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), vf[vf >= 0].astype(np.int32)


def test_device_growth_matches_reference_fixture(dev):
    """geobi_patch_grow from the fixture's seeds: the face lists of the reference's own mesh_get_neighbor_np, bit for bit;
    a ring-count-limited growth; an unlimited one."""
    from geobi_gnn_amd import meshprep, patches
    fx = load_fixture('patches_n8.npz')
    fv = torch.from_numpy(fx['faces']).to(dev).int().contiguous()
    V = fx['points'].shape[0]
    rowptr, lst = meshprep.vertex_faces(fv, V)
    vf32 = meshprep.vf_padded32(rowptr, lst, V)
    off = 0
    for seed, (nf, nv) in zip(fx['seeds'], fx['sizes']):
        got = patches.patch_grow(fv, vf32, int(seed), neighbor_count=int(fx['sub_size']))
        assert np.array_equal(got.cpu().numpy(), fx['select_faces'][off:off + nf])
        off += nf
    assert np.array_equal(patches.patch_grow(fv, vf32, 5, ring_count=2).cpu().numpy(), fx['ring2_from_face5'])
    full = patches.patch_grow(fv, vf32, 0).cpu().numpy()
    assert full.shape[0] == fv.shape[0] and np.unique(full).shape[0] == fv.shape[0]


@pytest.mark.parametrize('freq,sub', [(12, 500), (32, 3000), (45, 20000), (20, 100000)])
def test_device_split_equals_the_sequential_statement(dev, freq, sub):
    """The whole split (growth chain + seed selection on the device) against the sequential statement
    (oracle/mesh_ops.py) on meshes whose rings outgrow one thread's slot chunk and the LDS ring cache's first buffer;
    sub > F: one patch listing every face."""
    from geobi_gnn_amd import meshgen, meshprep, patches
    from oracle import mesh_ops
    noisy, _, faces = meshgen.noisy_icosphere(freq, 0.2, seed=11)
    pts = torch.as_tensor(noisy, dtype=torch.float32, device=dev)
    fv = torch.as_tensor(faces, dtype=torch.int32, device=dev).contiguous()
    V = pts.shape[0]
    got = [s.cpu().numpy() for s in patches.split_faces(pts, fv, sub)]
    # the oracle walks the same incidence order (ascending face ids per vertex) and the same fp32 d2
    rowptr, lst = _host_incidence(faces, V)
    centroid = pts.mean(0, keepdim=True)
    d2 = ((pts[fv.long()].mean(1) - centroid) ** 2).sum(1).cpu().numpy()
    want = mesh_ops.split_faces(d2, np.ascontiguousarray(faces, dtype=np.int32), rowptr, lst, sub)
    assert len(got) == len(want)
    for g, (_, w) in zip(got, want):
        assert np.array_equal(g, w)
    # a second split right behind the first (the mailbox and the side stream are reused)
    again = [s.cpu().numpy() for s in patches.split_faces(pts, fv, sub, ahead=3)]
    assert len(again) == len(got) and all(np.array_equal(a, g) for a, g in zip(again, got))


def _disk_mesh(k, rings):
    """A triangulated disk: one centre vertex of valence k, `rings` concentric rings of k points (exercises the wider
    incidence-row instantiations of the growth kernel)."""
    pts = [(0.0, 0.0, 0.0)]
    for r in range(1, rings + 1):
        for i in range(k):
            a = 2 * np.pi * (i + 0.5 * r) / k
            pts.append((r * np.cos(a), r * np.sin(a), 0.05 * np.sin(3 * a) * r))
    faces = []
    ring = lambda r, i: 1 + (r - 1) * k + (i % k)
    for i in range(k):
        faces.append((0, ring(1, i), ring(1, i + 1)))
    for r in range(1, rings):
        for i in range(k):
            faces.append((ring(r, i), ring(r + 1, i), ring(r, i + 1)))
            faces.append((ring(r, i + 1), ring(r + 1, i), ring(r + 1, i + 1)))
    return np.asarray(pts, dtype=np.float32), np.asarray(faces, dtype=np.int32)


@pytest.mark.parametrize('k,rings,sub', [(12, 6, 40), (20, 8, 150), (40, 5, 90)])
def test_device_split_on_wide_incidence_rows(dev, k, rings, sub):
    """Valence 12 / 20 / 40 at the centre: the 16-, 32- and 64-entry instantiations against the sequential statement."""
    from geobi_gnn_amd import meshprep, patches
    from oracle import mesh_ops
    pts_h, faces = _disk_mesh(k, rings)
    pts = torch.from_numpy(pts_h).to(dev)
    fv = torch.from_numpy(faces).to(dev).contiguous()
    V = pts.shape[0]
    rowptr, lst = meshprep.vertex_faces(fv, V)
    assert int(meshprep.vf_padded32(rowptr, lst, V).shape[1]) == k
    got = [s.cpu().numpy() for s in patches.split_faces(pts, fv, sub)]
    rp, ls = _host_incidence(faces, V)
    centroid = pts.mean(0, keepdim=True)
    d2 = ((pts[fv.long()].mean(1) - centroid) ** 2).sum(1).cpu().numpy()
    want = mesh_ops.split_faces(d2, faces, rp, ls, sub)
    assert len(got) == len(want)
    for g, (_, w) in zip(got, want):
        assert np.array_equal(g, w)


def test_device_split_beyond_the_lds_bitmaps(dev):
    """More faces / vertices than the kernel's LDS bitmaps hold (n = 115: 264 500 faces, 132 252 vertices): the
    instantiation with per-patch stamps in HBM."""
    from geobi_gnn_amd import meshgen, patches
    from oracle import mesh_ops
    noisy, _, faces = meshgen.noisy_icosphere(115, 0.2, seed=3)
    pts = torch.as_tensor(noisy, dtype=torch.float32, device=dev)
    fv = torch.as_tensor(faces, dtype=torch.int32, device=dev).contiguous()
    V = pts.shape[0]
    assert fv.shape[0] > 262144 and V > 131072
    got = [s.cpu().numpy() for s in patches.split_faces(pts, fv, 60000)]
    rp, ls = _host_incidence(faces, V)
    centroid = pts.mean(0, keepdim=True)
    d2 = ((pts[fv.long()].mean(1) - centroid) ** 2).sum(1).cpu().numpy()
    want = mesh_ops.split_faces(d2, np.ascontiguousarray(faces, dtype=np.int32), rp, ls, 60000)
    assert len(got) == len(want)
    for g, (_, w) in zip(got, want):
        assert np.array_equal(g, w)


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


def _c2_mesh(i):
    """Mesh i of the stand-in for dataset/Synthetic/test_list.txt (tools/test_synthetic.py)."""
    from geobi_gnn_amd import meshgen
    n, sg = (16, 22, 32, 45)[i % 4], (0.1, 0.2, 0.3)[i % 3]
    return n, meshgen.noisy_icosphere(n, sg, seed=100 + i)


@pytest.mark.parametrize('i', [3, 7])
def test_c2_full_size_patch_split(dev, i):
    """BASELINE configs[1], the meshes of the test list that exceed one patch: n = 45 (F = 40 500) split at the
    reference's sub_size = 20000 (test_dual.py:158).  Patches through the network 8 per pass or one by one give the
    same bits; the device merge equals a torch restatement of test_dual.py:49-61; the split is a cover of the mesh
    by patches of exactly sub_size faces (dataset.py:156-193)."""
    from geobi_gnn_amd import meshprep, network, patches
    from geobi_gnn_amd.infer import predict_one_submesh
    n, (noisy, clean, faces) = _c2_mesh(i)
    assert n == 45 and faces.shape[0] == 40500
    pts = torch.from_numpy(noisy).to(dev)
    fv = torch.from_numpy(faces).to(dev).int().contiguous()
    V, F = pts.shape[0], fv.shape[0]
    torch.manual_seed(0)
    net = network.DualGNN().to(dev).eval()
    out = patches.predict_mesh(net, pts, fv, sub_size=20000, n_iter=60, gt_points=clean)
    one = patches.predict_mesh(net, pts, fv, sub_size=20000, n_iter=60, patch_batch=1)
    assert out['n_patches'] == one['n_patches'] >= 3
    assert torch.equal(out['Np'], one['Np']) and torch.equal(out['Vp'], one['Vp'])
    assert torch.equal(out['V_updated'], one['V_updated'])

    parts = list(patches.split_patches(pts, fv, 20000))
    assert len(parts) == out['n_patches']
    covered = torch.zeros(F, dtype=torch.int32, device=dev)
    for sel, v_idx, f_sub in parts:
        assert sel.shape[0] == 20000 and int(torch.unique(sel).shape[0]) == 20000
        covered[sel.long()] += 1
        assert torch.equal(v_idx.long()[f_sub.long()], fv.long()[sel.long()])      # renumbering maps back
    assert int(covered.min()) >= 1 and int(covered.max()) > 1                      # a cover, with overlap

    rowptr, lst = meshprep.vertex_faces(fv, V)
    g_v = meshprep.ring_graph(0, fv, rowptr, lst, V)
    centroid = pts.mean(0, keepdim=True)
    scale = float((1.0 / meshprep.mean_edge_length(pts, g_v)).item())
    sum_v = torch.zeros((V, 1), device=dev)
    Vp = torch.zeros((V, 3), device=dev)
    Np = torch.zeros((F, 3), device=dev)
    for sel, v_idx, f_sub in parts:
        dual = meshprep.build_dual_data(pts[v_idx.long()], f_sub, device=dev, centroid=centroid, scale=scale)
        vert_p, norm_p = predict_one_submesh(net, dual)
        sum_v[v_idx.long()] += 1
        Vp[v_idx.long()] += vert_p
        Np[sel.long()] += norm_p
    Vp = Vp / sum_v / scale + centroid
    Np = torch.nn.functional.normalize(Np, dim=1)
    assert float((out['Vp'] - Vp).abs().max()) <= 1e-5 * float(Vp.abs().max())
    assert float((out['Np'] - Np).abs().max()) <= 1e-5
    assert torch.isfinite(out['V_updated']).all() and 0.0 <= out['angle1'] <= 180.0 and 0.0 <= out['angle2'] <= 180.0


@pytest.mark.parametrize('i', [0, 1])
def test_c2_single_patch_meshes_against_oracle(dev, i):
    """BASELINE configs[1], meshes that fit one patch (n = 16, 22): the whole inference chain of test_dual.py:24-87
    -- device preprocessing, network, de-normalisation, 60-sweep vertex update, both angular errors -- element-wise
    against the CPU oracle run on the reference-layout tensors with the HIP path's clusters replayed."""
    from geobi_gnn_amd import meshprep, network, patches
    from oracle import ref_model as R, pyg_ops as P
    from oracle.weights import make_state_dict
    from helpers import install_replay, rel_err
    n, (noisy, clean, faces) = _c2_mesh(i)
    sd = make_state_dict(R.DualGNN().state_dict(), 12 + i)
    net = network.DualGNN().to(dev)
    net.load_state_dict(sd)
    net.eval()
    out = patches.predict_mesh(net, noisy, faces, sub_size=20000, n_iter=60, gt_points=clean)
    assert out['n_patches'] == 1
    raw = []
    for m in (net.gnn_v.pooling1, net.gnn_v.pooling2, net.gnn_f.pooling1, net.gnn_f.pooling2):
        raw += [c.cpu() for c in m.last_clusters]
    # the oracle reads what the reference's loader would hand over (COO with self loops, loop weights included)
    dv, df = meshprep.build_dual_data(noisy, faces, device=dev, reference_layout=True)
    a = P.Data(dv.x.cpu(), dv.edge_index.cpu(), edge_weight=dv.edge_weight.cpu())
    b = P.Data(df.x.cpu(), df.edge_index.cpu(), edge_weight=df.edge_weight.cpu(), fv_indices=df.fv_indices.cpu())
    ora = R.DualGNN()
    ora.load_state_dict(sd)
    install_replay(ora, raw)
    with torch.no_grad():
        vo, no, _ = ora((a, b))
    cen, scale = dv.meta['centroid'].cpu(), dv.meta['scale']
    Vo = vo / scale + cen
    assert rel_err(out['Vp'].cpu(), Vo) < 1e-5
    assert float((out['Np'].cpu() - no).abs().max()) < 1e-5
    fv64 = torch.from_numpy(faces).long()
    Vu = R.update_position2(Vo, fv64, dv.meta['vf_indices'].cpu(), no, n_iter=60)
    assert rel_err(out['V_updated'].cpu(), Vu) < 2e-5
    Nt = R.computer_face_normal(torch.from_numpy(clean), fv64)
    assert abs(out['angle1'] - R.error_n(no, Nt).item()) < 1e-3
    assert abs(out['angle2'] - R.error_n(R.computer_face_normal(Vu, fv64), Nt).item()) < 1e-3


_MP_WORKER = r'''
import os, sys, time
T0 = time.time()
def stamp(what):          # phase times on stderr: a slow run says where it waited (rendezvous, GPU, reduction)
    print('[rank %%s] %%-22s +%%.1f s' %% (os.environ.get('RANK'), what, time.time() - T0), file=sys.stderr, flush=True)
import numpy as np, torch
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from geobi_gnn_amd import network, patches, meshgen
from geobi_gnn_amd.parallel import init_distributed
stamp('imports')
rank, world, device = init_distributed()
stamp('process group (gloo)')
assert world == 2 and device.type == 'cuda'
torch.manual_seed(0)
net = network.DualGNN().to(device).eval()
noisy, clean, faces = meshgen.noisy_icosphere(20, 0.2, seed=31)            # F = 8000 -> 5 patches of 2000 faces
stamp('network + mesh')
out = patches.predict_mesh(net, noisy, faces, sub_size=2000, n_iter=10, gt_points=clean)
stamp('sharded predict_mesh')
assert out['n_patches'] >= 4
if rank == 0:
    solo = patches.predict_mesh(net, noisy, faces, sub_size=2000, n_iter=10, gt_points=clean, distributed=False)
    assert solo['n_patches'] == out['n_patches']
    # each patch's prediction is the same bits on either rank; only the order of the merge additions differs
    assert float((solo['Vp'] - out['Vp']).abs().max()) <= 1e-5 * float(solo['Vp'].abs().max())
    assert float((solo['Np'] - out['Np']).abs().max()) <= 1e-5
    assert abs(solo['angle1'] - out['angle1']) < 1e-3 and abs(solo['angle2'] - out['angle2']) < 1e-3
else:
    assert out['Vp'] is None and out['V_updated'] is None
stamp('single-rank comparison')
dist.barrier()
dist.destroy_process_group()
stamp('done')
print('rank', rank, 'ok')
'''


def test_patch_scatter_two_ranks_on_one_device(dev, tmp_path):
    """SURVEY 8e inference sharding, rehearsed with 2 ranks sharing this box's one GPU (gloo): the ranks take
    alternate patches of one mesh, one reduction merges them on rank 0, result == the single-rank run."""
    import os, subprocess, sys, time
    from helpers import free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'mp_worker.py'
    script.write_text(_MP_WORKER % {'root': root})
    # gloo looks its interface up by host name, which need not resolve on a GPU box: pin it to the loopback
    env = dict(os.environ, GEOBI_ALL_RANKS_ON_DEVICE0='1', GEOBI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0',
               GLOO_SOCKET_IFNAME='lo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), str(script)]
    t0 = time.time()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    phases = '\n'.join(l for l in out.stderr.splitlines() if l.startswith('[rank '))
    print('two ranks on one device: %.1f s\n%s' % (took, phases))        # shown by pytest -rP / on failure
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count('ok') == 2
    assert took < 120, 'two-rank rehearsal took %.0f s:\n%s' % (took, phases)


def test_executor_on_open_patches_equals_module_path(dev):
    """Patches are open meshes (boundaries, irregular valences): the cases where a matching needs more than the default
    rounds or a coarse row exceeds the sort-free width are repaired inside the whole-network executor; the merged result
    is bit-identical to the module-by-module path and no call falls back."""
    from geobi_gnn_amd import network, patches, executor
    n, (noisy, clean, faces) = _c2_mesh(3)
    pts = torch.from_numpy(noisy).to(dev)
    fv = torch.from_numpy(faces).to(dev).int().contiguous()
    torch.manual_seed(0)
    net = network.DualGNN().to(dev).eval()
    was = executor.ENABLED
    try:
        executor.ENABLED = False
        ref = patches.predict_mesh(net, pts, fv, sub_size=6000, n_iter=5, patch_batch=3)
        executor.ENABLED = True
        before = dict(executor.STATS)
        out = patches.predict_mesh(net, pts, fv, sub_size=6000, n_iter=5, patch_batch=3)
        one = patches.predict_mesh(net, pts, fv, sub_size=6000, n_iter=5, patch_batch=1)
    finally:
        executor.ENABLED = was
    assert out['n_patches'] == ref['n_patches'] >= 7
    assert torch.equal(out['Np'], ref['Np']) and torch.equal(out['Vp'], ref['Vp'])
    assert torch.equal(one['Np'], ref['Np']) and torch.equal(one['Vp'], ref['Vp'])
    assert executor.STATS['calls'] - before['calls'] >= 3 + out['n_patches']
    assert executor.STATS['fallback'] == before['fallback']


def test_meshes_in_flight_together_equal_one_by_one(dev):
    """patches.predict_many: two and three host threads, each with its own stream (= its own context of the library),
    run whole-mesh inference incl. the patch split at the same time; every mesh gets the bits it gets on its own.
    (Round 4 found the look-back scan's state zeroed on the null stream, which a thread starting later raced with.)"""
    from geobi_gnn_amd import meshgen, network, patches
    torch.manual_seed(1)
    net = network.DualGNN().to(dev).eval()
    meshes = []
    for i, n in enumerate((10, 16, 22, 14, 22, 12)):
        noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=60 + i)
        meshes.append((torch.as_tensor(noisy, dtype=torch.float32, device=dev),
                       torch.as_tensor(faces, dtype=torch.int32, device=dev), None))
    want = [patches.predict_mesh(net, p, f, sub_size=3000, n_iter=5) for p, f, _ in meshes]
    assert max(w['n_patches'] for w in want) >= 3
    for workers in (2, 3):
        for _ in range(3):
            got = patches.predict_many(net, meshes, workers=workers, sub_size=3000, n_iter=5)
            for g, w in zip(got, want):
                assert g['n_patches'] == w['n_patches']
                assert torch.equal(g['Np'], w['Np']) and torch.equal(g['V_updated'], w['V_updated'])


def test_predict_batch_equals_mesh_by_mesh(dev):
    """patches.predict_batch: the small meshes of a list go through the network and the vertex update as one disjoint-union
    mesh (a group is closed when it would pass max_faces), the large ones through the patch split; every mesh must get the
    bits predict_mesh gives it alone -- predictions, updated vertices and both angular errors."""
    from geobi_gnn_amd import network, meshgen, patches
    torch.manual_seed(5)
    net = network.DualGNN().to(dev).eval()
    meshes = []
    for i, n in enumerate((8, 10, 14, 9, 12, 8, 11)):
        noisy, clean, faces = meshgen.noisy_icosphere(n, (0.1, 0.2, 0.3)[i % 3], seed=600 + i)
        meshes.append((torch.as_tensor(noisy, dtype=torch.float32, device=dev),
                       torch.as_tensor(faces, dtype=torch.int32, device=dev),
                       torch.as_tensor(clean, dtype=torch.float32, device=dev)))
    with torch.no_grad():
        want = [patches.predict_mesh(net, p, f, sub_size=3000, n_iter=10, gt_points=g) for p, f, g in meshes]
        got = patches.predict_batch(net, meshes, max_faces=6000, sub_size=3000, n_iter=10)
        one = patches.predict_batch(net, meshes[:1], sub_size=3000, n_iter=10)
    assert [w['n_patches'] for w in want] == [g['n_patches'] for g in got]
    assert want[2]['n_patches'] > 1                       # n = 14: 3 920 faces, split
    for w, g in zip(want + want[:1], got + one):
        for k in ('Vp', 'Np', 'V_updated'):
            assert torch.equal(w[k], g[k]), k
        assert w['angle1'] == g['angle1'] and w['angle2'] == g['angle2']


@pytest.mark.parametrize('split_group,patch_batch', [(2, 5), (3, 2), (2, 1)])
def test_predict_batch_split_meshes_in_groups(dev, split_group, patch_batch):
    """patches.predict_batch with several meshes ABOVE the patch size: their growth chains run side by side, a network pass
    takes patches of more than one mesh (patch_batch 2 cuts a mesh's patches across passes), one vertex update runs over the
    union of the meshes -- and every mesh still gets predict_mesh's bits."""
    from geobi_gnn_amd import network, meshgen, patches
    torch.manual_seed(6)
    net = network.DualGNN().to(dev).eval()
    meshes = []
    for i, n in enumerate((14, 9, 16, 13, 15, 8)):
        noisy, clean, faces = meshgen.noisy_icosphere(n, (0.1, 0.2, 0.3)[i % 3], seed=700 + i)
        meshes.append((torch.as_tensor(noisy, dtype=torch.float32, device=dev),
                       torch.as_tensor(faces, dtype=torch.int32, device=dev),
                       torch.as_tensor(clean, dtype=torch.float32, device=dev)))
    with torch.no_grad():
        want = [patches.predict_mesh(net, p, f, sub_size=3000, n_iter=10, gt_points=g, patch_batch=patch_batch) for p, f, g in meshes]
        got = patches.predict_batch(net, meshes, max_faces=6000, sub_size=3000, n_iter=10, patch_batch=patch_batch,
                                    split_group=split_group)
    assert [w['n_patches'] for w in want] == [g['n_patches'] for g in got]
    assert sum(w['n_patches'] > 1 for w in want) == 4
    for w, g in zip(want, got):
        for k in ('Vp', 'Np', 'V_updated'):
            assert torch.equal(w[k], g[k]), k
        assert w['angle1'] == g['angle1'] and w['angle2'] == g['angle2']


def test_predict_batch_edge_cases(dev):
    """An empty list, meshes without ground truth (no angles, no host read for them), a single mesh, a group cut by
    max_faces smaller than any mesh (every small mesh then goes alone) and numpy inputs."""
    from geobi_gnn_amd import network, meshgen, patches
    torch.manual_seed(7)
    net = network.DualGNN().to(dev).eval()
    assert patches.predict_batch(net, []) == []
    raw = [meshgen.noisy_icosphere(n, 0.2, seed=800 + i) for i, n in enumerate((8, 9, 10))]
    with torch.no_grad():
        want = [patches.predict_mesh(net, noisy, faces, sub_size=3000, n_iter=5) for noisy, _, faces in raw]
        got = patches.predict_batch(net, [(noisy, faces) for noisy, _, faces in raw], sub_size=3000, n_iter=5)     # numpy in
        alone = patches.predict_batch(net, [(noisy, faces) for noisy, _, faces in raw], max_faces=1, sub_size=3000, n_iter=5)
    for w, g, a in zip(want, got, alone):
        assert g['angle1'] is None and g['angle2'] is None and g['n_patches'] == 1
        for k in ('Vp', 'Np', 'V_updated'):
            assert torch.equal(w[k], g[k]) and torch.equal(w[k], a[k]), k
    bad = (raw[0][0], raw[0][2].copy())
    bad[1][0, 0] = 10 ** 6
    with pytest.raises(Exception):
        patches.predict_batch(net, [bad], sub_size=3000, n_iter=5)


def test_one_face_patch(dev):
    """A connected component of ONE face becomes a patch of its own (the growth stops when a component is exhausted): its
    facet graph has no edges.  The graph fill used to be called with an empty column array and refused it
    (tools/fuzz_mesh.py found it on icospheres with half of their faces removed)."""
    from geobi_gnn_amd import network, meshgen, patches
    noisy, _, faces = meshgen.noisy_icosphere(6, 0.2, seed=3)
    V = noisy.shape[0]
    lone = np.array([[10.0, 10.0, 10.0], [10.5, 10.0, 10.0], [10.0, 10.5, 10.1]], dtype=noisy.dtype)
    pts = np.concatenate([noisy, lone], 0)
    fcs = np.concatenate([faces, [[V, V + 1, V + 2]]], 0)
    torch.manual_seed(2)
    net = network.DualGNN().to(dev).eval()
    with torch.no_grad():
        r = patches.predict_mesh(net, pts, fcs, sub_size=300, n_iter=5)
        alone = patches.predict_mesh(net, lone, np.array([[0, 1, 2]]), sub_size=300, n_iter=5)
    assert r['n_patches'] >= 3
    for out in (r, alone):
        assert bool(torch.isfinite(out['Vp']).all() and torch.isfinite(out['Np']).all() and torch.isfinite(out['V_updated']).all())
        assert bool(((out['Np'].norm(dim=1) - 1).abs() < 1e-4).all())
