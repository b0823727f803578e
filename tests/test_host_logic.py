"""CPU tests of the host side: C-ABI surface, input contract of the generator, data-parallel
plumbing over gloo (world_size 2).  No compute call reaches the HIP library here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from geobi_gnn_amd import _lib
    protos = _lib.parse_header()
    # every prototype in the public header, found independently of the binding's parser
    src = open(_lib.HEADER).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    names = set(re.findall(r'\b(geobi_\w+)\s*\(', src))
    assert names == set(protos), names ^ set(protos)
    assert len(names) >= 30
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(['bash', os.path.join(ROOT, 'geobi_gnn_amd', 'csrc', 'build.sh')])
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    lib = _lib.lib()
    assert lib.geobi_version() >= 100
    # pure host queries work without a GPU
    assert lib.geobi_feast_ldz(6) == 56 and lib.geobi_feast_ldz(64) == 576
    assert lib.geobi_feast_bwd_ws_bytes(1000, 7000, 64, 32) > 0


def test_product_refuses_cpu_tensors_and_never_imports_oracle():
    from geobi_gnn_amd import network, meshgen, _lib
    net = network.DualGNN()
    dv, df = meshgen.synthetic_dual_data(2, 0.2, 0)
    with pytest.raises(_lib.GeobiError):
        net((dv, df))
    for root, _, files in os.walk(os.path.join(ROOT, 'geobi_gnn_amd')):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f
    text = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'cpu_baseline' in text


def test_state_dict_keys_match_reference_layout():
    from geobi_gnn_amd import network
    net = network.DualGNN()
    keys = list(net.state_dict().keys())
    assert keys[0] == 'gnn_v.l_conv1.c' or 'gnn_v.l_conv1.lin.weight' in keys
    assert {'gnn_v.l_conv1.lin.weight', 'gnn_v.l_conv1.u.weight', 'gnn_v.l_conv1.c', 'gnn_v.l_conv1.bias',
            'gnn_f.r_conv4.lin.weight', 'fc_v1.weight', 'fc_v2.bias', 'fc_f1.weight', 'fc_f2.bias'} <= set(keys)
    assert sum(p.numel() for p in net.parameters()) == 939128
    assert sum(p.numel() for p in network.DualGNN(force_depth=True).parameters()) == 937078
    assert net.gnn_v.l_conv1.lin.weight.shape == (9 * 32, 6) and net.gnn_f.l_conv1.u.weight.shape == (9, 12)
    # same keys as the oracle restatement (which loads the reference's own state_dict)
    from oracle import ref_model as R
    assert keys == list(R.DualGNN().state_dict().keys())


@pytest.mark.parametrize('n', [1, 3, 11])
def test_synthetic_mesh_contract(n):
    from geobi_gnn_amd import meshgen
    dv, df = meshgen.synthetic_dual_data(n, 0.2, seed=n)
    F, V = 20 * n * n, 10 * n * n + 2
    assert dv.x.shape == (V, 6) and df.x.shape == (F, 6) and df.fv_indices.shape == (F, 3)
    assert dv.edge_index.shape[1] == 3 * F + V
    if n >= 3:
        assert df.edge_index.shape[1] == 13 * F - 60
    # vertex graph: sorted symmetric pairs, then V self loops appended (dataset.py:211-213)
    ei = dv.edge_index
    assert torch.equal(ei[:, -V:], torch.arange(V).repeat(2, 1))
    body = ei[:, :-V]
    key = body[0] * V + body[1]
    assert bool((key[1:] > key[:-1]).all())
    assert set(key.tolist()) == set((body[1] * V + body[0]).tolist())
    # facet graph: row-major sorted, self loops inline (data_util.py:436-456)
    kf = df.edge_index[0] * F + df.edge_index[1]
    assert bool((kf[1:] > kf[:-1]).all()) and int((df.edge_index[0] == df.edge_index[1]).sum()) == F
    assert bool((dv.edge_weight > 0).all()) and bool(torch.isfinite(df.edge_weight).all())
    # unit normals, unit-mean-edge-length scaling
    assert torch.allclose(dv.x[:, 3:].norm(dim=1), torch.ones(V), atol=1e-5)
    assert torch.allclose(df.y.norm(dim=1), torch.ones(F), atol=1e-5)


def test_union_batch_and_batched_losses():
    from geobi_gnn_amd import meshgen
    from geobi_gnn_amd.data import union_batch
    from geobi_gnn_amd.parallel import batched_losses
    a = meshgen.synthetic_dual_data(2, 0.2, 0)
    b = meshgen.synthetic_dual_data(3, 0.3, 1)
    dv, df = union_batch([a, b])
    Va, Fa = a[0].x.shape[0], a[1].x.shape[0]
    assert dv.x.shape[0] == Va + b[0].x.shape[0] and dv.mesh_ptr.tolist() == [0, Va, dv.x.shape[0]]
    assert int(dv.edge_index[:, :a[0].edge_index.shape[1]].max()) < Va
    assert int(dv.edge_index[:, a[0].edge_index.shape[1]:].min()) >= Va
    assert int(df.fv_indices[Fa:].min()) >= Va
    g = torch.Generator().manual_seed(0)
    vp = torch.randn(dv.x.shape[0], 3, generator=g)
    npred = torch.randn(df.x.shape[0], 3, generator=g)
    lv, ln = batched_losses(vp, npred, dv, df)
    want_v = 0.5 * ((vp[:Va] - a[0].y).abs().sum(1).mean() + (vp[Va:] - b[0].y).abs().sum(1).mean())
    want_n = 0.5 * ((npred[:Fa] - a[1].y).abs().sum(1).mean() + (npred[Fa:] - b[1].y).abs().sum(1).mean())
    assert torch.allclose(lv, want_v, rtol=1e-5) and torch.allclose(ln, want_n, rtol=1e-5)
    sc = dv.shallow_copy()
    sc.x = None
    assert dv.x is not None and sc.edge_index is dv.edge_index


def test_shard_indices_partition():
    from geobi_gnn_amd.parallel import shard_indices
    for world in (1, 2, 3, 8):
        parts = [shard_indices(29, r, world, seed=5, epoch=2) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(29))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_indices(10, 0, 2, seed=1, epoch=0) != shard_indices(10, 0, 2, seed=1, epoch=1)


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from geobi_gnn_amd import network
from geobi_gnn_amd.parallel import init_distributed, GradBucket, reduce_sums, shard_indices
rank, world, device = init_distributed('gloo')
assert world == 2 and device.type == 'cpu'
torch.manual_seed(0)
net = network.DualGNN()                       # parameters only; no HIP call is made on CPU
bucket = GradBucket(net.parameters())
assert bucket.flat.numel() == 939128
bucket.zero()
for i, p in enumerate(net.parameters()):
    p.grad.add_(float(rank + 1) * (i + 1))    # what backward would accumulate in place
flat = bucket.all_reduce_mean()
for i, p in enumerate(net.parameters()):
    assert torch.allclose(p.grad, torch.full_like(p.grad, 1.5 * (i + 1))), i
# the facet half first, then the vertex half (what runs under the vertex branch's backward on the MI355X): the same bucket
g0 = torch.Generator().manual_seed(17 + rank)
mine = torch.randn(bucket.flat.numel(), generator=g0)
bucket.flat.copy_(mine)
one_shot = bucket.all_reduce_mean().clone()
bucket.flat.copy_(mine)
off = bucket.facet_offset(net)
assert 0 < off < bucket.flat.numel() and off == sum(p.numel() for k, p in net.named_parameters()
                                                      if k.startswith(('gnn_v.', 'fc_v')))
assert torch.equal(bucket.all_reduce_mean_split(off), one_shot)
for i, p in enumerate(net.parameters()):
    p.grad.fill_(1.5 * (i + 1))
# the optimiser sees the reduced gradients through the same views
opt = torch.optim.SGD(net.parameters(), lr=1.0)
before = net.fc_f2.bias.detach().clone()
opt.step()
idx = [i for i, (k, _) in enumerate(net.named_parameters()) if k == 'fc_f2.bias'][0]
assert torch.allclose(net.fc_f2.bias, before - 1.5 * (idx + 1))
# flat parameters: one optimiser tensor aliasing every module parameter
from geobi_gnn_amd.parallel import FlatParameters
net2 = network.DualGNN()
sd0 = {k: v.clone() for k, v in net2.state_dict().items()}
fp = FlatParameters(net2)
assert all(torch.equal(v, sd0[k]) for k, v in net2.state_dict().items())
fp.bucket.zero()
net2.fc_v1.bias.grad.add_(1.0)
torch.optim.SGD(fp.parameters(), lr=0.5).step()
assert torch.allclose(net2.fc_v1.bias, sd0['fc_v1.bias'] - 0.5) and torch.equal(net2.fc_v2.bias, sd0['fc_v2.bias'])
sums = reduce_sums([rank + 1.0, 10.0], device)
assert sums == [3.0, 20.0]
mine = shard_indices(7, rank, world, seed=3)
gathered = [None, None]
dist.all_gather_object(gathered, mine)
assert sorted(gathered[0] + gathered[1]) == list(range(7))
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_data_parallel_gloo_world2(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % {'root': ROOT})
    from helpers import free_port
    port = str(free_port())
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, OMP_NUM_THREADS='2',
               CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='', GLOO_SOCKET_IFNAME='lo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', port, str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count('ok') == 2


_PATCH_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from geobi_gnn_amd.parallel import init_distributed, owns_patch, reduce_patch_sums, rank_world
rank, world, device = init_distributed('gloo')
assert (rank, world) == rank_world() and world == 2 and device.type == 'cpu'
fx = np.load(os.path.join(%(root)r, 'tests', 'golden', 'patches_n8.npz'))
V, F = fx['points'].shape[0], fx['faces'].shape[0]

def accumulate(keep):
    """test_dual.py:53-58 over the fixture's patches with a stand-in prediction per patch (the network itself
    only runs on the MI355X); what geobi_patch_accumulate does, in torch."""
    Vp, Np, cnt = torch.zeros(V, 3), torch.zeros(F, 3), torch.zeros(V, dtype=torch.int32)
    fo = vo = 0
    for k, (nf, nv) in enumerate(fx['sizes']):
        sel = torch.from_numpy(fx['select_faces'][fo:fo + nf]).long()
        v_idx = torch.from_numpy(fx['v_idx'][vo:vo + nv]).long()
        fo += nf; vo += nv
        if not keep(k):
            continue
        g = torch.Generator().manual_seed(k)
        Vp[v_idx] += torch.randn(nv, 3, generator=g)
        Np[sel] += torch.randn(nf, 3, generator=g)
        cnt[v_idx] += 1
    return Vp, Np, cnt

mine = accumulate(lambda k: owns_patch(k, rank, world))
owned = [k for k in range(len(fx['sizes'])) if owns_patch(k, rank, world)]
both = [None, None]
dist.all_gather_object(both, owned)
assert sorted(both[0] + both[1]) == list(range(len(fx['sizes']))) and not set(both[0]) & set(both[1])
reduce_patch_sums(list(mine), dst=0)
if rank == 0:
    full = accumulate(lambda k: True)
    assert torch.equal(mine[2], full[2]) and int(full[2].min()) >= 1
    assert torch.allclose(mine[0], full[0], atol=1e-6) and torch.allclose(mine[1], full[1], atol=1e-6)
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_patch_scatter_gloo_world2(tmp_path):
    """Multi-GPU inference: the patches of one mesh dealt over 2 ranks and reduced onto rank 0 give the
    single-rank sums (partition is exact and disjoint; visit counts bit-equal)."""
    script = tmp_path / 'patch_worker.py'
    script.write_text(_PATCH_WORKER % {'root': ROOT})
    from helpers import free_port
    port = str(free_port())
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, OMP_NUM_THREADS='2',
               CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='', GLOO_SOCKET_IFNAME='lo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', port, str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count('ok') == 2


def test_op_tape_matches_autograd_on_a_dag():
    """ops.Tape (the reverse-mode tape DualGNN runs its ~70 ops through) against torch.autograd on a
    small DAG with a tensor used twice, a two-output-gradient op and a branch without gradient."""
    from torch.autograd import Function
    from geobi_gnn_amd import ops

    class Scale(Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x * w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            return g * w, (g * x).sum(0, keepdim=True) if ctx.needs_input_grad[1] else None

    class AddPair(Function):
        @staticmethod
        def forward(ctx, a, b, k):
            ctx.k = k
            return a + k * b

        @staticmethod
        def backward(ctx, g):
            return g, ctx.k * g, None

    torch.manual_seed(0)
    x = torch.randn(5, 3)
    w1 = torch.randn(1, 3, requires_grad=True)
    w2 = torch.randn(1, 3, requires_grad=True)
    const = torch.randn(5, 3)

    def graph(apply):
        h = apply(Scale, x, w1)              # x needs no grad
        a = apply(Scale, h, w2)              # h used twice below
        b = apply(AddPair, h, a, 2.0)
        dead = apply(Scale, const, const[:1])   # no gradient flows here
        return apply(AddPair, b, h, -0.5), dead

    out_ref, _ = graph(lambda fn, *a: fn.apply(*a))
    g = torch.randn_like(out_ref)
    out_ref.backward(g)
    ref = (w1.grad.clone(), w2.grad.clone())

    tape = ops.Tape(record=True)
    with torch.no_grad():
        out_t, dead = graph(tape.apply)
    assert torch.equal(out_t, out_ref.detach())
    assert len(tape.nodes) == 4                                   # the dead branch was not recorded
    leaf = tape.backward({id(out_t): g})
    assert torch.allclose(leaf[id(w1)], ref[0], atol=1e-6) and torch.allclose(leaf[id(w2)], ref[1], atol=1e-6)
    assert tape.nodes == []
    # a tape that does not record (inference) keeps nothing alive
    t2 = ops.Tape(record=False)
    with torch.no_grad():
        graph(t2.apply)
    assert t2.nodes == []
    # apply_op falls back to torch.autograd when no tape is active and uses the active one otherwise
    assert ops._ACTIVE_TAPE[0] is None
    with ops.use_tape(tape):
        assert ops._ACTIVE_TAPE[0] is tape
    assert ops._ACTIVE_TAPE[0] is None


def test_patch_growth_matches_reference_fixture():
    """The sequential statement of the ring growth (oracle/oracle_c.c: oracle_patch_grow, the spec of the device kernel
    geobi_patch_grow) against face lists produced by the reference's own data_util.mesh_get_neighbor_np
    (tests/golden/patches_n8.npz, oracle/gen_golden.py:gen_patches), and the seed loop of dataset.py:156-193 around it."""
    from geobi_gnn_amd import meshgen
    from oracle import mesh_ops
    from helpers import load_fixture
    fx = load_fixture('patches_n8.npz')
    faces = fx['faces']
    V = fx['points'].shape[0]
    vf = meshgen.vertex_faces(faces.astype(np.int64), V)
    counts = (vf >= 0).sum(1)
    rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    lst = vf[vf >= 0].astype(np.int32)                       # row-major: per vertex, in row order
    fv = np.ascontiguousarray(faces, dtype=np.int32)
    off = 0
    for seed, (nf, nv) in zip(fx['seeds'], fx['sizes']):
        got = mesh_ops.patch_grow(fv, rowptr, lst, int(seed), neighbor_count=int(fx['sub_size']))
        assert np.array_equal(got, fx['select_faces'][off:off + nf])
        off += nf
    assert np.array_equal(mesh_ops.patch_grow(fv, rowptr, lst, 5, ring_count=2), fx['ring2_from_face5'])
    # unlimited growth reaches every face of the (connected) sphere exactly once
    full = mesh_ops.patch_grow(fv, rowptr, lst, 0)
    assert full.shape[0] == faces.shape[0] and np.unique(full).shape[0] == faces.shape[0]
    # the seed loop: same seeds, same patches as the reference's split
    pts = fx['points'].astype(np.float32)
    d2 = ((pts[faces].mean(1) - pts.mean(0, keepdims=True)) ** 2).sum(1)
    split = mesh_ops.split_faces(d2, fv, rowptr, lst, int(fx['sub_size']))
    assert [s for s, _ in split] == [int(s) for s in fx['seeds']]
    assert np.array_equal(np.concatenate([f for _, f in split]), fx['select_faces'])


def test_processed_file_round_trip(tmp_path):
    """The processed-mesh cache (counterpart of dataset.py:153,182,276) loads with weights_only=True."""
    from geobi_gnn_amd import meshgen
    from geobi_gnn_amd.data import save_processed, load_processed
    dv, df = meshgen.synthetic_dual_data(3, 0.2, seed=2)
    path = str(tmp_path / 'ico3.pt')
    save_processed((dv, df), path)
    lv, lf = load_processed(path)
    for a, b in ((dv, lv), (df, lf)):
        assert set(a.keys()) == set(b.keys())
        for k in a.keys():
            va, vb = getattr(a, k), getattr(b, k)
            if torch.is_tensor(va):
                assert torch.equal(va, vb), k
            elif isinstance(va, dict):
                assert set(va) == set(vb) and all(torch.equal(va[m], vb[m]) if torch.is_tensor(va[m]) else va[m] == vb[m]
                                                  for m in va)
            else:
                assert va == vb, k
    assert lv.num_nodes == dv.num_nodes and lf.fv_indices.dtype == torch.int64


def test_random_rotate_is_a_rigid_motion_of_both_graphs():
    """dataset.py:39-69: positions, normals and targets of both graphs turn by the same rotation."""
    from geobi_gnn_amd import meshgen
    from geobi_gnn_amd.data import RandomRotate
    dv, df = meshgen.synthetic_dual_data(3, 0.2, seed=4)
    ref = (dv.clone(), df.clone())
    for z_only in (True, False):
        dv, df = ref[0].clone(), ref[1].clone()
        rot = RandomRotate(z_rotated=z_only, rng=np.random.default_rng(5))
        m = torch.from_numpy(RandomRotate(z_rotated=z_only, rng=np.random.default_rng(5)).matrix()).float()
        assert torch.allclose(m @ m.t(), torch.eye(3), atol=1e-6) and abs(float(torch.det(m)) - 1) < 1e-6
        rot((dv, df))
        for a, b in ((dv, ref[0]), (df, ref[1])):
            assert torch.allclose(a.x[:, :3], b.x[:, :3] @ m, atol=1e-5)
            assert torch.allclose(a.x[:, 3:6], b.x[:, 3:6] @ m, atol=1e-6)
            assert torch.allclose(a.y, b.y @ m, atol=1e-5)
        # lengths and normal lengths are preserved; z-only rotations keep z
        assert torch.allclose(dv.x[:, :3].norm(dim=1), ref[0].x[:, :3].norm(dim=1), atol=1e-4)
        if z_only:
            assert torch.allclose(dv.x[:, 2], ref[0].x[:, 2], atol=1e-6)


def _run_bench(*argv, **env_extra):
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=env, text=True,
                          capture_output=True, timeout=240)


def test_bench_launches_its_own_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` without WORLD_SIZE: the parent starts torch.distributed.run as a child, the two
    ranks rendezvous over gloo, rank 0's single JSON line comes back through the parent."""
    import json
    r = _run_bench('--gpus', '2', '--plumbing')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rank_sum'] == 3.0 and out['backend'] == 'gloo'
    assert 'torch.distributed.run' in r.stderr and '--nproc-per-node 2' in r.stderr
    # the fields that let a scaling line explain itself, produced by the helpers of the real line:
    # backend + world size, the all-reduce alone, every rank's own step time
    cfg = out['config']
    assert cfg['backend'] == 'gloo' and cfg['world_size'] == 2
    assert cfg['collective_us'] > 0 and 0 < cfg['rank_ms_per_step']['min'] <= cfg['rank_ms_per_step']['max']
    # three successive mean all-reduces of (1, 2): 1.5 every time
    assert cfg['bucket_mean'] == 1.5


def test_bench_relays_a_failing_child_and_checks_world_size():
    # no GPU here: the ranks refuse to run the workload; the parent must exit non-zero and print no result line
    r = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '0')
    assert r.returncode != 0 and r.stdout.strip() == ''
    # launched by an outer torchrun with a different world size: refused, not silently mis-reported
    r = _run_bench('--gpus', '2', '--steps', '1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    assert r.returncode != 0 and 'WORLD_SIZE=1 but --gpus 2' in r.stderr


def _opt(**kw):
    import argparse
    from geobi_gnn_amd import train_util
    base = train_util.add_training_flags(argparse.ArgumentParser()).parse_args([])
    for k, v in kw.items():
        setattr(base, k, v)
    return base


def test_training_flags_match_the_reference_defaults():
    """same flag names and defaults as train_dual.py:57-82."""
    o = _opt()
    assert (o.loss_v, o.loss_n, o.loss_v_scale, o.loss_n_scale, o.wei_param) == ('L1', 'L1', 1, 1, 2)
    assert (o.max_epoch, o.batch_size, o.lr_sch, o.lr, o.lr_step, o.lr_decay) == (1000, 1, 'lmd', 0.001, [10], 1)
    assert (o.optimizer, o.momentum, o.beta1, o.beta2, o.weight_decay) == ('adam', 0.9, 0.9, 0.999, 0)


@pytest.mark.parametrize('name', ['step', 'multi_step', 'exp', 'auto', 'lmd'])
def test_lr_schedules_reproduce_the_reference_step_for_step(name):
    """The five schedules of train_dual.py:169-180, epoch by epoch, against the constructors written out as the
    reference writes them (incl. the plateau schedule stepping on the evaluation normal error, :261-264)."""
    from torch.optim import lr_scheduler
    from geobi_gnn_amd import train_util
    opt = _opt(lr_sch=name, lr=0.01, lr_decay=0.7, lr_step=[3, 7, 12] if name == 'multi_step' else [3])
    w = [torch.nn.Parameter(torch.ones(3)) for _ in range(2)]
    ours_opt, ref_opt = torch.optim.Adam([w[0]], lr=opt.lr), torch.optim.Adam([w[1]], lr=opt.lr)
    ours = train_util.make_scheduler(opt, ours_opt)
    if name == 'step':
        ref = lr_scheduler.StepLR(ref_opt, step_size=opt.lr_step[0], gamma=opt.lr_decay)
    elif name == 'multi_step':
        ref = lr_scheduler.MultiStepLR(ref_opt, milestones=opt.lr_step, gamma=opt.lr_decay)
    elif name == 'exp':
        ref = lr_scheduler.ExponentialLR(ref_opt, gamma=opt.lr_decay)
    elif name == 'auto':
        ref = lr_scheduler.ReduceLROnPlateau(ref_opt, factor=opt.lr_decay, patience=opt.lr_step[0])
    else:
        ref = lr_scheduler.LambdaLR(ref_opt, lr_lambda=lambda step: opt.lr_decay ** (step / opt.lr_step[0]))
    metric = [5.0, 4.0, 4.1, 4.2, 4.3, 4.4, 4.5, 3.0, 3.1, 3.2, 3.3, 3.4, 3.5, 3.6, 3.7, 3.8, 2.0, 2.1, 2.2, 2.3]
    seen = []
    for epoch in range(20):
        ours_opt.step(); ref_opt.step()
        train_util.step_scheduler(opt, ours, metric[epoch])
        if name == 'auto':
            ref.step(metric[epoch])
        else:
            ref.step()
        assert ours_opt.param_groups[0]['lr'] == ref_opt.param_groups[0]['lr'], (name, epoch)
        seen.append(ours_opt.param_groups[0]['lr'])
    assert len(set(seen)) > 2                       # the schedule actually moved the rate


def test_optimizers_eval_means_and_checkpoint_keys(tmp_path):
    """train_dual.py:162-167 optimiser settings, :246-259 node-count-weighted evaluation means, :270-276 best
    checkpoint -- whose file the reference's `net.load_state_dict(torch.load(path))` must accept (checked against
    the oracle's restatement of the reference module tree, strict key match)."""
    from geobi_gnn_amd import network, train_util
    from oracle import ref_model as R
    p = [torch.nn.Parameter(torch.ones(2))]
    a = train_util.make_optimizer(_opt(optimizer='adam', beta1=0.8, beta2=0.95, weight_decay=0.01), p)
    assert isinstance(a, torch.optim.Adam) and a.defaults['betas'] == (0.8, 0.95) and a.defaults['weight_decay'] == 0.01
    s = train_util.make_optimizer(_opt(optimizer='sgd', momentum=0.7), p)
    assert isinstance(s, torch.optim.SGD) and s.defaults['momentum'] == 0.7
    r = train_util.make_optimizer(_opt(optimizer='rmsprop'), p)
    assert isinstance(r, torch.optim.RMSprop) and r.defaults['alpha'] == 0.9
    with pytest.raises(ValueError):
        train_util.make_optimizer(_opt(optimizer='lbfgs'), p)

    meter = train_util.EvalMeter()
    rows = [(0.5, 0.2, 0.05, 3.0, 100, 196), (0.3, 0.1, 0.02, 1.0, 1000, 1996), (0.9, 0.4, 0.09, 7.5, 10, 16)]
    lv = lf = ev = ef = cv = cf = 0.0
    for l_v, l_f, e_v, e_f, nv, nf in rows:
        meter.add(l_v, l_f, e_v, e_f, nv, nf)
        lv += l_v * nv; lf += l_f * nf; ev += e_v * nv; ef += e_f * nf; cv += nv; cf += nf     # :246-251
    res = meter.result()
    assert res == {'eval_loss_v': lv / cv, 'eval_loss_f': lf / cf, 'eval_error_v': ev / cv, 'eval_error_f': ef / cf}

    net = network.DualGNN()
    path = str(tmp_path / 'best.pt')
    ck = train_util.BestCheckpoint(path)
    assert ck.update(net, 3.0) and not ck.update(net, 3.5) and ck.update(net, 2.5) and ck.best == 2.5
    sd = torch.load(path, map_location='cpu', weights_only=True)
    ora = R.DualGNN()
    assert list(sd.keys()) == list(ora.state_dict().keys())
    ora.load_state_dict(sd, strict=True)
    assert torch.equal(ora.fc_f2.weight, net.fc_f2.weight.detach())


def test_struct_mirrors_match_the_library_layout():
    """The ctypes mirrors of the header's structs have the size the library was compiled with (a drift would corrupt
    every call that passes one; geobi_abi_sizeof)."""
    import ctypes
    from geobi_gnn_amd import _lib as L, executor, data
    lib = L.lib()
    mirrors = {0: executor._Params, 1: executor._Level0, 2: executor._Out, 3: executor._Group, 4: data._CopySeg}
    for which, cls in mirrors.items():
        assert lib.geobi_abi_sizeof(which) == ctypes.sizeof(cls), (which, cls.__name__)
    assert lib.geobi_abi_sizeof(99) == 0


def test_union_batch_graphs_refuses_other_dtypes():
    """a float64 / int64 part must raise in the collate step, not be copied as 4-byte words."""
    import pytest
    from geobi_gnn_amd.data import _Concat
    cc = _Concat(torch.device('cpu'))
    with pytest.raises(ValueError):
        cc.cat([torch.zeros(4, dtype=torch.float64)])
    with pytest.raises(ValueError):
        cc.cat([torch.zeros(4, dtype=torch.int64)])


def test_sizes_beyond_the_documented_limits_are_rejected():
    """include/geobi_hip.h: GEOBI_MAX_NODES / GEOBI_MAX_EDGES.  An entry point returns an error for a larger call before
    anything touches the device (so this runs without a GPU) -- never a truncated 32-bit index inside a kernel."""
    import ctypes
    from geobi_gnn_amd import _lib as L
    lib = L.lib()
    max_nodes, max_edges = (1 << 24) - 1, (1 << 28) - 1
    one = ctypes.c_void_p(256)                     # never dereferenced: the size check comes first
    rc = lib.geobi_feast_fwd(one, None, 32, 0, max_nodes + 1, 10, one, one, one, one, one, one, 32, 1.0, one, one, None, None,
                             one, 0, None)
    assert rc != 0 and b'GEOBI_MAX_NODES' in lib.geobi_last_error()
    rc = lib.geobi_feast_fwd(one, None, 32, 0, 10, max_edges + 1, one, one, one, one, one, one, 32, 1.0, one, one, None, None,
                             one, 0, None)
    assert rc != 0 and b'GEOBI_MAX_EDGES' in lib.geobi_last_error()
    rc = lib.geobi_segment_sum(one, 32, one, one, max_nodes + 1, 0, one, None)
    assert rc != 0 and b'GEOBI_MAX_NODES' in lib.geobi_last_error()
    rc = lib.geobi_edge_weight_t10(one, 32, one, one, one, -1, one, None)
    assert rc != 0 and b'negative' in lib.geobi_last_error()

    class _L0(ctypes.Structure):
        _fields_ = [('N', ctypes.c_int64), ('E', ctypes.c_int64), ('rowptr', ctypes.c_void_p), ('col', ctypes.c_void_p),
                    ('row', ctypes.c_void_p), ('weight', ctypes.c_void_p)]
    from geobi_gnn_amd import executor
    big, ok = _L0(max_nodes + 1, 10, 256, 256, 256, 256), _L0(10, 10, 256, 256, 256, 256)
    prm, out = executor._Params(), executor._Out()
    rc = lib.geobi_net_forward(ctypes.byref(prm), ctypes.byref(big), ctypes.byref(ok), one, one, one, None, one, 1 << 20,
                               ctypes.byref(out), None)
    assert rc != 0 and b'GEOBI_MAX_NODES' in lib.geobi_last_error()


def test_flat_adam_refuses_options_it_does_not_implement():
    """a state dict of Adam(amsgrad=True) / Adam(maximize=True) loads into FlatAdam (same layout) but must not
    train with another update rule silently."""
    import pytest
    from geobi_gnn_amd.train_util import FlatAdam
    for kw in ({'amsgrad': True}, {'maximize': True}):
        p = torch.nn.Parameter(torch.zeros(4))
        donor = torch.optim.Adam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3, **kw)
        opt = FlatAdam([p], lr=1e-3)
        opt.load_state_dict(donor.state_dict())
        p.grad = torch.ones(4)
        with pytest.raises(ValueError, match=list(kw)[0]):
            opt.step()


def test_tensorboard_event_writer_round_trip(tmp_path):
    """train_util.SummaryWriter stands in for tensorboardX (train_dual.py:134-136,222-226; not installed here, nor is
    TensorBoard: the format is pinned by the CRC-32C known answer of RFC 3720, the TFRecord framing constants and the
    module's own reader, which checks both CRCs of every record)."""
    import struct
    from geobi_gnn_amd import train_util as T
    assert T._crc32c(b'123456789') == 0xE3069283
    assert T._crc32c(bytes(32)) == 0x8A9136AA              # RFC 3720 B.4: 32 bytes of zeros
    w = T.SummaryWriter(str(tmp_path / 'train'))
    want = []
    for it in range(1, 40):
        for tag, v in (('loss_v', 0.5 / it), ('error_f', 30.0 / it), ('dual_loss', -1.25e-3 * it)):
            w.add_scalar(tag, v, it * 4)
            want.append((it * 4, tag, struct.unpack('<f', struct.pack('<f', v))[0]))
    w.add_text('train_params', 'Namespace(lr=0.001)')
    w.close()
    assert T.read_scalars(w.path) == want
    assert (tmp_path / 'train' / 'train_params.txt').read_text() == 'Namespace(lr=0.001)'
    raw = open(w.path, 'rb').read()
    n, = struct.unpack('<Q', raw[:8])
    assert raw[12:12 + n].endswith(b'brain.Event:2')        # the file-version record comes first
    broken = bytearray(raw); broken[40] ^= 1
    bad = tmp_path / 'broken'
    bad.write_bytes(bytes(broken))
    with pytest.raises(ValueError):
        T.read_scalars(str(bad))


def test_learned_arena_size_is_monotone_and_quantised():
    """executor._learned_size: what later training passes on a level-0 shape allocate.  Needs that differ by a fraction of a
    percent from pass to pass (other meshes of the same shape) must map to ONE size -- a size that followed every pass made
    the block allocator hand out a fresh multi-GB block whenever the request grew past the cached one."""
    from geobi_gnn_amd import executor
    table, key = {}, (1, 2, 3, 4)
    q = 64 << 20
    sizes = [executor._learned_size(table, key, used) for used in (2674517504, 2675022592, 2674494720, 2673000000)]
    assert len(set(sizes)) == 1 and sizes[0] % q == 0 and sizes[0] >= int(1.05 * 2675022592)
    grown = executor._learned_size(table, key, 3 * 2674517504)
    assert grown > sizes[0] and grown % q == 0
    assert executor._learned_size(table, key, 1000) == grown                  # never shrinks
    assert executor._learned_size({}, key, 0) == q                            # a floor of one quantum
