"""Mesh groups in flight together (geobi_net_train_groups / executor.TrainGroups): the iterations of the reference's
gradient-accumulation loop (/root/reference/code/train_dual.py:199-218) overlapped on several streams.

Bars: per-mesh predictions BIT-identical to the single-union step (the union is per-mesh independent); the summed
gradient within 1e-5 of the union's (of each tensor's max: the sums over nodes are formed in another order); group
order A|B and B|A to the same bits; the losses add up to the step's losses."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _setup(dev, freq=12, n_mesh=4, force_depth=False):
    from geobi_gnn_amd import network, meshgen
    from geobi_gnn_amd.parallel import FlatParameters
    torch.manual_seed(3)
    net = network.DualGNN(force_depth=force_depth).to(dev)
    flat = FlatParameters(net)
    pairs = [meshgen.synthetic_dual_data(freq + 2 * (i % 2), (0.1, 0.2, 0.3)[i % 3], seed=40 + i) for i in range(n_mesh)]
    if force_depth:
        for dv, _ in pairs:
            d = dv.x[:, :3]
            dv.depth_direction = torch.nn.functional.normalize(d, dim=1)
    return net, flat.bucket, pairs


def _union(pairs, idx, dev):
    from geobi_gnn_amd.data import union_batch
    dv, df = union_batch([pairs[i] for i in idx]) if len(idx) > 1 else pairs[idx[0]]
    return dv.to(dev), df.to(dev)


def _union_step(net, bucket, dv, df):
    from geobi_gnn_amd.parallel import batched_losses
    bucket.zero()
    vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy()))
    lv, ln = batched_losses(vp, npred, dv, df, 'L1', 'L1')
    (lv + ln).backward()
    torch.cuda.synchronize()
    return vp.detach().clone(), npred.detach().clone(), float(lv), float(ln), bucket.flat.clone()


def _per_param_err(bucket, g, g_ref):
    worst, off = 0.0, 0
    for p in bucket.params:
        n = p.numel()
        a, b = g[off:off + n].double(), g_ref[off:off + n].double()
        worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-30)))
        off += n
    return worst


@pytest.mark.parametrize('split', [[[0, 1, 2, 3]], [[0, 1], [2, 3]], [[0], [1, 2], [3]], [[0], [1], [2], [3]]])
def test_groups_equal_the_union_step(dev, split):
    from geobi_gnn_amd import executor
    net, bucket, pairs = _setup(dev)
    dv, df = _union(pairs, [0, 1, 2, 3], dev)
    vp, npred, lv, ln, g_ref = _union_step(net, bucket, dv, df)
    tg = executor.TrainGroups(net, bucket).set_groups([_union(pairs, ix, dev) for ix in split])
    for _ in range(2):                                   # the second step runs on learned arena sizes
        losses = tg.step()
        torch.cuda.synchronize()
    assert tg.sequential_steps == 0
    tot = losses.sum(0).tolist()
    assert abs(tot[0] - lv) < 2e-6 * abs(lv) and abs(tot[1] - ln) < 2e-6 * abs(ln)
    tol = 1e-3 if any(len(ix) < 4 for ix in split) else 0.0
    # u.weight gradients of the deep layers are sums of cancelling terms: the existing bar of the model tests applies
    off = 0
    names = [n for n, _ in net.named_parameters()]
    for name, p in zip(names, bucket.params):
        n = p.numel()
        a, b = bucket.flat[off:off + n].double(), g_ref[off:off + n].double()
        e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        bar = 0.0 if len(split) == 1 else (1e-3 if name.endswith('.u.weight') else 1e-5)
        assert e <= bar, (name, e)
        off += n
    # per-mesh predictions: bit-identical to the union's rows
    ptr_v, ptr_f = dv.mesh_ptr.tolist(), df.mesh_ptr.tolist()
    for k, ix in enumerate(split):
        verts, normals = tg.prediction(k)
        assert torch.equal(verts, vp[ptr_v[ix[0]]:ptr_v[ix[-1] + 1]])
        assert torch.equal(normals, npred[ptr_f[ix[0]]:ptr_f[ix[-1] + 1]])


def test_groups_with_mean_pooling(dev):
    """pool_type='mean' (net_util.py:131-134) through the grouped step: same bars as the max-pooling network."""
    from geobi_gnn_amd import executor, network, meshgen
    from geobi_gnn_amd.parallel import FlatParameters
    torch.manual_seed(9)
    net = network.DualGNN(pool_type='mean').to(dev)
    bucket = FlatParameters(net).bucket
    pairs = [meshgen.synthetic_dual_data(10 + 2 * (i % 2), 0.2, seed=80 + i) for i in range(4)]
    dv, df = _union(pairs, [0, 1, 2, 3], dev)
    vp, npred, lv, ln, g_ref = _union_step(net, bucket, dv, df)
    tg = executor.TrainGroups(net, bucket).set_groups([_union(pairs, [0, 1], dev), _union(pairs, [2, 3], dev)])
    losses = tg.step(); torch.cuda.synchronize()
    assert tg.sequential_steps == 0
    tot = losses.sum(0).tolist()
    assert abs(tot[0] - lv) < 2e-6 * abs(lv) and abs(tot[1] - ln) < 2e-6 * abs(ln)
    assert float((bucket.flat - g_ref).abs().max() / g_ref.abs().max()) < 1e-5
    verts, normals = tg.prediction(0)
    assert torch.equal(verts, vp[:verts.shape[0]]) and torch.equal(normals, npred[:normals.shape[0]])


def test_group_order_does_not_change_a_bit(dev):
    from geobi_gnn_amd import executor
    net, bucket, pairs = _setup(dev)
    a, b = _union(pairs, [0, 1], dev), _union(pairs, [2, 3], dev)
    tg = executor.TrainGroups(net, bucket).set_groups([a, b])
    tg.step(); torch.cuda.synchronize()
    g_ab, l_ab = bucket.flat.clone(), tg.losses.clone()
    tg.step(); torch.cuda.synchronize()
    assert torch.equal(bucket.flat, g_ab)                # run to run
    tg.set_groups([b, a])
    tg.step(); torch.cuda.synchronize()
    assert torch.equal(bucket.flat, g_ab)
    assert torch.equal(tg.losses, l_ab.flip(0))


def test_groups_with_depth_direction_and_l2(dev):
    from geobi_gnn_amd import executor, network
    from geobi_gnn_amd.parallel import batched_losses
    net, bucket, pairs = _setup(dev, force_depth=True)
    dv, df = _union(pairs, [0, 1, 2, 3], dev)
    bucket.zero()
    vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy()))
    lv, ln = batched_losses(vp, npred, dv, df, 'L2', 'L2')
    network.dual_loss(lv, ln, v_scale=2.0, n_scale=0.5).backward()
    torch.cuda.synchronize()
    g_ref = bucket.flat.clone()
    tg = executor.TrainGroups(net, bucket, 'L2', 'L2', v_scale=2.0, n_scale=0.5)
    tg.set_groups([_union(pairs, [0, 1], dev), _union(pairs, [2, 3], dev)])
    losses = tg.step(); torch.cuda.synchronize()
    assert tg.sequential_steps == 0
    tot = losses.sum(0).tolist()
    assert abs(tot[0] - 2.0 * float(lv)) < 1e-5 * abs(float(lv)) and abs(tot[1] - 0.5 * float(ln)) < 1e-5 * abs(float(ln))
    assert float((bucket.flat - g_ref).abs().max() / g_ref.abs().max()) < 1e-5


def test_groups_outside_the_fast_path_run_group_by_group(dev):
    """A matching that needs more rounds than the cap makes the executor decline: the step then runs group by group
    through the network's autograd node and gives the same gradient."""
    from geobi_gnn_amd import executor, _lib as L
    net, bucket, pairs = _setup(dev)
    groups = [_union(pairs, [0, 1], dev), _union(pairs, [2, 3], dev)]
    tg = executor.TrainGroups(net, bucket).set_groups(groups)
    tg.step(); torch.cuda.synchronize()
    g_fast = bucket.flat.clone()
    net.gnn_v.pooling1.pool_step = 3                      # outside the executor's fast path
    try:
        tg2 = executor.TrainGroups(net, bucket).set_groups(groups)
        tg2.step(); torch.cuda.synchronize()
        assert tg2.sequential_steps == 1
    finally:
        net.gnn_v.pooling1.pool_step = 2
    # (a different pooling depth is a different network: only check that the path ran and produced finite numbers)
    assert bool(torch.isfinite(bucket.flat).all())
    # same network through the sequential path: equal to the grouped step within the gradient bar
    tg3 = executor.TrainGroups(net, bucket).set_groups(groups)
    tg3._sequential(); torch.cuda.synchronize()
    assert float((bucket.flat - g_fast).abs().max() / g_fast.abs().max()) < 1e-5


def test_groups_grow_their_arenas(dev):
    from geobi_gnn_amd import executor
    net, bucket, pairs = _setup(dev)
    groups = [_union(pairs, [0, 1], dev), _union(pairs, [2, 3], dev)]
    tg = executor.TrainGroups(net, bucket).set_groups(groups)
    tg.step(); torch.cuda.synchronize()
    g_ref = bucket.flat.clone()
    before = executor.STATS['arena_retry']
    tg2 = executor.TrainGroups(net, bucket).set_groups(groups)
    for pr in tg2._prep:
        pr['res']['arena'] = torch.empty(1 << 20, dtype=torch.uint8, device=dev)      # far too small
        executor._LEARNED_GROUP[pr['shape']] = 1 << 20
    tg2.step(); torch.cuda.synchronize()
    assert executor.STATS['arena_retry'] > before
    assert torch.equal(bucket.flat, g_ref)


def test_two_host_threads_drive_the_library_at_once(dev):
    """Per-context state of the library is per host thread (side streams, events, scan state, size mailbox): two Python
    threads, each with its own stream, run whole-network inference passes at the same time and get the numbers a single
    thread gets."""
    from geobi_gnn_amd import network, meshgen
    torch.manual_seed(5)
    net = network.DualGNN().to(dev)
    meshes = [tuple(d.to(dev) for d in meshgen.synthetic_dual_data(10 + 2 * i, 0.2, seed=70 + i)) for i in range(2)]

    def run(k, out, reps):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            for _ in range(reps):
                dv, df = meshes[k]
                with torch.no_grad():
                    vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy()))
            s.synchronize()
            out[k] = (vp.clone(), npred.clone())
    ref = {}
    for k in range(2):
        run(k, ref, 1)
    got, errs = {}, []

    def guarded(k):
        try:
            run(k, got, 6)
        except Exception as e:                            # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=guarded, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert torch.equal(got[k][0], ref[k][0]) and torch.equal(got[k][1], ref[k][1])


def test_backward_marks_where_the_facet_half_is_final(dev):
    """geobi_net_backward_facet_events: the backward records the two events once the facet branch's gradients are enqueued;
    reducing the bucket in two halves behind them (parallel.GradBucket.all_reduce_mean_split) leaves the same gradient."""
    from geobi_gnn_amd import executor
    from geobi_gnn_amd.parallel import batched_losses
    net, bucket, pairs = _setup(dev)
    dv, df = _union(pairs, [0, 1], dev)
    _, _, _, _, g_ref = _union_step(net, bucket, dv, df)
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    for e in evs:
        e.record()
    torch.cuda.synchronize()
    off = bucket.facet_offset(net)
    assert off == sum(p.numel() for k, p in net.named_parameters() if k.startswith(('gnn_v.', 'fc_v')))
    bucket.zero()
    vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy()))
    lv, ln = batched_losses(vp, npred, dv, df, 'L1', 'L1')
    executor.FACET_EVENTS = evs
    try:
        (lv + ln).backward()
    finally:
        executor.FACET_EVENTS = None
    # a stream that only waits for the two events sees the facet half final
    comm = torch.cuda.Stream(device=dev)
    for e in evs:
        comm.wait_event(e)
    with torch.cuda.stream(comm):
        tail = bucket.flat[off:].clone()
    comm.synchronize()
    assert torch.equal(tail, g_ref[off:])
    torch.cuda.synchronize()
    assert torch.equal(bucket.flat, g_ref)
    assert bucket.all_reduce_mean_split(off, evs, comm) is bucket.flat          # world size 1: nothing to reduce
