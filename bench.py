"""Benchmark of the hot path: DualGNN forward + loss + backward (+ optimiser step) on synthetic meshes.

Contract (driver):  python bench.py --gpus N --steps K --warmup W     (N > 1: launched through
torch.distributed.run, one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[2] "Synthetic train, batch=4 meshes (~20k faces each), fwd+bwd":
per rank and step, 4 noisy icospheres of frequency 32 (F = 20 480, V = 10 242; 337 862 level-0
edges each, self loops included, as the dataset delivers them) processed as one disjoint-union
graph with per-mesh mean losses (= the reference's gradient accumulation over batch_size = 4
single-mesh steps, train_dual.py:211-218), L1/L1 losses, Adam lr 1e-3.  Weak scaling: every rank
owns a different batch; the only collective is the all-reduce of the 3.76 MB gradient bucket.
Timed region: forward, loss, backward, gradient all-reduce, optimiser step.  Inputs are resident
in HBM before the clock starts.

metric  M-edges/s = level-0 edge_index columns of all meshes of all ranks / wall seconds.
roofline  the forward fused FeaSt kernel (aggregation -- the "scatter-add" of the path -- with the node
          transform fused behind it), its dominant instantiation: ALGORITHMIC bytes (SURVEY.md 8d,
          B_agg with the fused write term 4 N C_out) / launch time measured with HIP events on the
          launch stream in a separate pass; `traffic` / `frac_by_counters` quote the committed PMC passes.
cpu_baseline  the PyG-shaped CPU oracle (same op decomposition as the reference) on ONE mesh of
          the same size, all host cores, rank 0 at N = 1 only.
extra     rank 0 at N = 1, after the timed region (never part of `value`):
          infer        BASELINE configs[1] (one mesh n = 32) and configs[3] (one unsplit scan n = 87): network ms,
                       M-edges/s and the fused kernel's algorithmic fraction of the HBM peak (test_dual.py:18-22,44-87)
          fresh_batch  ms/step when every step unions 4 OTHER pre-processed meshes from a pool of 12 (what a loader
                       hands over, train_dual.py:199-201): the union is built inside each step (two launches:
                       every array is a shifted concatenation of per-mesh arrays); plus the one-off per-mesh build
          test_list    BASELINE configs[1] as the reference runs it (test_dual.py:90-148): the 29-mesh stand-in list through
                       patches.predict_mesh (preprocessing, patch split at 20 000 faces, network, merge, vertex update):
                       total ms, meshes/s, M-edges/s, number of patch-split meshes, share of time outside the network
          mesh_groups  the timed step with the batch as 2 mesh groups in flight together (geobi_net_train_groups)
N > 1     `config.collective_us` (the gradient all-reduce alone, HIP events), `config.rank_ms_per_step` (min / max).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA (v_mfma_f32_32x32x2_f32), same table
FREQ = 32                  # icosphere frequency: F = 20 480 faces
BATCH = 4


def make_batch(rank, device, freq=FREQ, batch=BATCH, groups=1):
    """-> (data_v, data_f) of the 4-mesh union, edges per step, and the same meshes as `groups` unions of batch / groups
    meshes each (mesh order kept: group k holds meshes k * batch / groups ...)."""
    from geobi_gnn_amd import meshgen
    from geobi_gnn_amd.data import union_batch
    pairs = []
    for i in range(batch):
        sigma = (0.1, 0.2, 0.3)[i % 3]                                   # the _n1/_n2/_n3 noise levels
        pairs.append(meshgen.synthetic_dual_data(freq, sigma, seed=200 + rank * batch + i))
    edges = sum(p[0].edge_index.shape[1] + p[1].edge_index.shape[1] for p in pairs)
    dv, df = union_batch(pairs)
    per = batch // groups
    parts = []
    for k in range(groups):
        sub = pairs[k * per:(k + 1) * per]
        a, b = union_batch(sub) if len(sub) > 1 else sub[0]
        parts.append((a.to(device), b.to(device)))
    return dv.to(device), df.to(device), edges, parts


def grouped_step(tg, bucket, opt, collective=True):
    """The same optimiser step with the batch's meshes as mesh groups in flight together (executor.TrainGroups: the
    iterations of the reference's accumulation loop, train_dual.py:199-218, overlapped; gradients summed in group order)."""
    losses = tg.step()
    if collective:
        bucket.all_reduce_mean()
    opt.step()
    return losses


def train_step(net, bucket, opt, dv0, df0, collective=True):
    from geobi_gnn_amd import network
    from geobi_gnn_amd.parallel import batched_losses
    bucket.zero()
    dv, df = dv0.shallow_copy(), df0.shallow_copy()      # the forward rewrites .x on the bag it gets
    vp, npred, _ = net((dv, df))
    lv, ln = batched_losses(vp, npred, dv0, df0, 'L1', 'L1')
    loss = network.dual_loss(lv, ln)
    split = collective and SPLIT_ALLREDUCE is not None
    if split:
        from geobi_gnn_amd import executor
        executor.FACET_EVENTS = SPLIT_ALLREDUCE['events']
    loss.backward()
    if split:
        executor.FACET_EVENTS = None
        bucket.all_reduce_mean_split(SPLIT_ALLREDUCE['offset'], SPLIT_ALLREDUCE['events'], SPLIT_ALLREDUCE['stream'])
    elif collective:
        bucket.all_reduce_mean()
    opt.step()
    return loss


# GEOBI_SPLIT_ALLREDUCE=1 at N > 1: the facet half of the bucket is all-reduced under the vertex branch's backward
# (parallel.GradBucket.all_reduce_mean_split); built and checked for equality with the one-shot form (gloo tests), NOT
# measured -- RCCL only runs in the driver's multi-GPU tier.  Off by default.
SPLIT_ALLREDUCE = None


def enable_split_allreduce(net, bucket, device):
    global SPLIT_ALLREDUCE
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    for e in evs:
        e.record()                              # creates the underlying hipEvent_t (torch does so lazily)
    SPLIT_ALLREDUCE = {'events': evs, 'offset': bucket.facet_offset(net), 'stream': torch.cuda.Stream(device=device)}


def measure_roofline(net, bucket, opt, dv, df, steps=6):
    """Separate pass: every forward launch of the fused FeaSt kernel (aggregation = the path's "scatter-add", with the
    node transform fused behind it through LDS) bracketed by HIP events on its stream.  ALGORITHMIC bytes per launch =
    SURVEY.md 8d's B_agg with the fused write term W = 4 N C_out (csrc/feast_fused.hip: feast_fused_bytes)."""
    from geobi_gnn_amd import _lib as L
    lib = L.lib()
    lib.geobi_prof_enable(1)
    for _ in range(steps):
        train_step(net, bucket, opt, dv, df, collective=False)      # rank-local pass: no collective
    torch.cuda.synchronize()
    best = None
    total = {'launches': 0, 'ms': 0.0, 'bytes': 0.0}
    for cin in (6, 12, 32, 64, 128):
        for cout in (32, 64, 128):
            n, ms, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
            L.check(lib.geobi_prof_collect(cin * 1000 + cout, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)),
                    'prof_collect')
            if n.value == 0:
                continue
            total['launches'] += n.value; total['ms'] += ms.value; total['bytes'] += by.value
            if best is None or ms.value > best['ms']:
                best = {'cin': cin, 'cout': cout, 'launches': n.value, 'ms': ms.value, 'bytes': by.value}
    lib.geobi_prof_enable(0)
    ach = best['bytes'] / (best['ms'] * 1e-3) / 1e9
    lc = best['cin'] if best['cin'] in (6, 12) else 0
    rows = 32 if os.environ.get('GEOBI_TILE16', '1') == '0' else 16
    kname = 'feast_fused_kernel<%d,0,%d,%d,%d>' % (best['cin'], lc, best['cout'] // 32, rows)   # <C, MODE, LC, NT, ROWS>
    avg_us = best['ms'] * 1e3 / best['launches']
    traffic, traffic_src = pmc_traffic(kname)
    # the same kernel against its other ceiling: the node transform on the fp32 matrix cores (2 N 9 Cin Cout flops per
    # launch; the dominant layer runs on both level-0 graphs, N averaged over its launches)
    n_avg = (dv.x.shape[0] + df.x.shape[0]) / 2.0
    tflops = 2.0 * n_avg * 9 * best['cin'] * best['cout'] / (avg_us * 1e-6) / 1e12
    # ... and the aggregation beside it: 2 x 9 Cin flops per (edge + self loop) as packed FMAs.  fp32 MFMA and VALU work do
    # not overlap on a gfx950 SIMD (profiles/r03_overlap_probe.txt): the two shares of the kernel meet ONE 64 flop / cycle
    # / SIMD ceiling, numerically the fp32 MFMA peak
    e_avg = delivered_edges(dv, df) / 2.0
    tflops_all = tflops + 2.0 * e_avg * 9 * best['cin'] / (avg_us * 1e-6) / 1e12
    out = {
        'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(ach / HBM_PEAK_GBS, 4),
        # HBM bytes per launch by the PMC counters: QUOTED from the committed rocprofv3 --pmc passes over this same
        # workload (counters cannot be read from inside the process), not measured in this run
        'traffic': traffic, 'traffic_source': traffic_src,
        'frac_by_counters': None if traffic is None else round(traffic / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
        'kernel': kname, 'layer': 'FeaStConv %d -> %d' % (best['cin'], best['cout']),
        'mfma_side': {'achieved': round(tflops, 1), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                      'frac': round(tflops / MFMA_F32_PEAK_TFLOPS, 4),
                      'note': 'valid when the dominant layer is a level-0 layer (64 -> 32 on the bench workload)'},
        'fp32_side': {'achieved': round(tflops_all, 1), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                      'frac': round(tflops_all / MFMA_F32_PEAK_TFLOPS, 4),
                      'note': 'node transform (matrix cores) + aggregation (packed FMAs) against the one fp32 ceiling of a '
                              'SIMD: the two do not overlap on gfx950 (profiles/r03_overlap_probe.txt)'},
        'launches': best['launches'], 'avg_us': round(avg_us, 2),
        'timing': 'HIP events bound to each launch of the kernel (hipExtLaunchKernelGGL start / stop events on its stream): the '
                  'dispatch\'s own execution time, what rocprofv3 reports for it; a pair of hipEventRecord packets around a launch '
                  'reads ~4.7 us more (rounds 1-3 and the first sets of round 4)',
        'alg_bytes_per_launch': round(best['bytes'] / best['launches']),
        'all_instantiations': {'launches': total['launches'],
                               'achieved': round(total['bytes'] / (total['ms'] * 1e-3) / 1e9, 1),
                               'avg_us': round(total['ms'] * 1e3 / total['launches'], 2)},
    }
    return out


def measure_mfma(net, bucket, opt, dv, df, steps=3):
    """Second separate pass: the node-level GEMMs (z Wf, g Wf^T, r' W: gemm_nn; weight gradients:
    gemm_tn) bracketed by HIP events, with the side stream off so that each is timed alone.
    BASELINE.json's north_star asks for the MFMA utilisation of the node GEMM beside the HBM line."""
    from geobi_gnn_amd import _lib as L
    lib = L.lib()
    lib.geobi_set_overlap(0)
    lib.geobi_prof_enable(4)
    for _ in range(steps):
        train_step(net, bucket, opt, dv, df, collective=False)
    torch.cuda.synchronize()
    out = {'bound': 'mfma', 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'pmc': PMC_MFMA + ' (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F32 of every kernel that runs a node '
                  'transform, one rocprofv3 --pmc pass over this workload; tools/pmc_mfma.py): quoted below, the '
                  'launch-bracketed figures of this run beside them',
           'by_counters': pmc_mfma()}
    for name, tag in (('gemm_nn', 1), ('gemm_tn', 2)):
        n, ms, fl = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        L.check(lib.geobi_prof_collect(tag, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), 'prof_collect')
        if n.value:
            tf = fl.value / (ms.value * 1e-3) / 1e12
            out[name] = {'launches_per_step': n.value // steps, 'avg_us': round(ms.value * 1e3 / n.value, 2),
                         'gflop_per_step': round(fl.value / steps / 1e9, 2), 'achieved': round(tf, 1),
                         'frac': round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
    lib.geobi_prof_enable(0)
    lib.geobi_set_overlap(1)
    return out


def delivered_edges(dv, df):
    """Level-0 edge_index columns as the dataset delivers them (dataset.py:212-213,232: self loops appended to the
    vertex graph, inline in the facet graph) for a pair whose adjacency is the loop-free CSR of the device
    preprocessing: one loop per node on top of the stored edges."""
    return dv.graph().E + dv.x.shape[0] + df.graph().E + df.x.shape[0]


def fused_kernel_stats(lib, run, reps=3):
    """HIP-event time and algorithmic bytes of the forward fused FeaSt launches of `run()`, dominant instantiation."""
    from geobi_gnn_amd import _lib as L
    lib.geobi_prof_enable(1)
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    best = None
    for cin in (6, 12, 32, 64, 128):
        for cout in (32, 64, 128):
            n, ms, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
            L.check(lib.geobi_prof_collect(cin * 1000 + cout, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)),
                    'prof_collect')
            if n.value and (best is None or ms.value > best[1]):
                best = ('%d->%d' % (cin, cout), ms.value, by.value, n.value)
    lib.geobi_prof_enable(0)
    gbs = best[2] / (best[1] * 1e-3) / 1e9
    return {'layer': 'FeaStConv ' + best[0], 'launches': best[3], 'avg_us': round(best[1] * 1e3 / best[3], 2),
            'achieved': round(gbs, 1), 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4)}


def measure_infer(net, device):
    """BASELINE configs[1] / configs[3]: single-mesh inference (test_dual.py:18-22: forward under no_grad) of one
    noisy icosphere n = 32 (F = 20 480) and one unsplit scan-sized n = 87 (F = 151 380), inputs resident in HBM.
    Same weights as the training net (random init + the timed steps).  The meshes come from the device
    preprocessing (meshprep.build_dual_data: same fields as the host generator, loop-free CSR)."""
    from geobi_gnn_amd import meshgen, meshprep, infer, _lib as L
    lib = L.lib()
    out = {}
    for key, name, n, reps in (('configs[1]', 'Synthetic test_list single-mesh inference', 32, 30),
                               ('configs[3]', 'Kinect_Fusion-sized large scan, unsplit', 87, 15)):
        noisy, clean, faces = meshgen.noisy_icosphere(n, 0.2, seed=7)
        dv, df = meshprep.build_dual_data(noisy, faces, clean, device=device)
        edges = delivered_edges(dv, df)
        vf = dv.meta['vf_indices']
        run = lambda: infer.predict_one_submesh(net, (dv, df))
        for _ in range(5):
            run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize(); t_net = (time.perf_counter() - t0) / reps
        full = lambda: infer.predict_one(net, dv, df, dv.meta['centroid'], dv.meta['scale'], vf, 60)
        full()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(max(3, reps // 3)):
            full()
        torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / max(3, reps // 3)
        out[key] = {'workload': '%s: icosphere n=%d (F=%d, %d level-0 edges)' % (name, n, df.x.shape[0], edges),
                    'network_ms': round(t_net * 1e3, 3), 'M_edges_per_s': round(edges / t_net / 1e6, 1),
                    'with_60_sweep_vertex_update_ms': round(t_all * 1e3, 3),
                    'fused_feast_kernel': fused_kernel_stats(lib, run)}
        log('infer %s: %.3f ms network, %.1f M-edges/s' % (key, t_net * 1e3, edges / t_net / 1e6))
    return out


def measure_fresh_batch(net, bucket, opt, device, freq, pool_size=12, steps=12, warm=3):
    """The same training step when every step sees 4 OTHER meshes (train_dual.py:199-201: the loader hands a new
    mesh each iteration): a pool of pre-processed meshes resident in HBM (device preprocessing, timed as the one-off
    per-mesh structure build), each step unions 4 of them (data.union_batch_graphs: CSR concatenation, no sort) and
    rebuilds what the replayed bench batch caches -- reverse-edge index, vertex -> corner lists, loss weights."""
    from geobi_gnn_amd import meshgen, meshprep
    from geobi_gnn_amd.data import union_batch_graphs
    raw = [meshgen.noisy_icosphere(freq, (0.1, 0.2, 0.3)[i % 3], seed=500 + i) for i in range(pool_size)]
    from geobi_gnn_amd.network import _fv_index

    def build(noisy, clean, faces):
        # everything that depends on ONE mesh only, once per mesh: graphs, normals, weights, features (device
        # preprocessing), the reverse-edge index of both graphs and the vertex -> corner lists of the face table
        dv, df = meshprep.build_dual_data(noisy, faces, clean, device=device)
        dv.graph().ensure_in(); df.graph().ensure_in()
        _fv_index(df, dv.x.shape[0])[1].get()
        return dv, df
    build(raw[0][0], raw[0][1], raw[0][2])                                             # warm the kernels' first launch
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pool = [build(noisy, clean, faces) for noisy, clean, faces in raw]
    torch.cuda.synchronize(); build_ms = (time.perf_counter() - t0) * 1e3 / pool_size
    edges = [delivered_edges(dv, df) for dv, df in pool]

    def step(k):
        idx = [(k + 3 * i) % pool_size for i in range(BATCH)]        # 12 different unions for k = 0 .. 11
        dv, df = union_batch_graphs([pool[i] for i in idx])
        train_step(net, bucket, opt, dv, df, collective=False)
        return sum(edges[i] for i in idx)
    for k in range(warm):
        step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    total = 0
    for k in range(steps):
        total += step(warm + k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    log('fresh batch: %.3f ms/step, per-mesh structure build %.3f ms' % (dt / steps * 1e3, build_ms))
    return {'workload': 'every step unions %d other meshes (n=%d) from a pool of %d pre-processed meshes resident in '
                        'HBM; the union (every array incl. reverse-edge index, corner lists, loss weights as shifted '
                        'concatenations of the per-mesh ones: two launches) inside the step' % (BATCH, freq, pool_size),
            'ms_per_step': round(dt / steps * 1e3, 3), 'M_edges_per_s': round(total / dt / 1e6, 2), 'steps': steps,
            'structure_build_ms_per_mesh': round(build_ms, 3),
            'structure_build': 'device preprocessing of one raw mesh (points + faces -> both level-0 CSR graphs, '
                               'normals, bilateral weights, features; meshprep.build_dual_data) + its reverse-edge '
                               'indices and vertex -> corner lists, one-off per mesh'}


def measure_test_list(net, device, sub_size=20000):
    """BASELINE configs[1] as the reference runs it (test_dual.py:90-148, predict_dir): EVERY mesh of the test list
    through the whole chain -- preprocessing, patch split above `sub_size` faces, network, overlap merge, 60-sweep vertex
    update, the two angular errors.  The Synthetic test set is an external download: SURVEY 8d's stand-in list, 29 noisy
    icospheres (one per name of dataset/Synthetic/test_list.txt), n from {16, 22, 32, 45} x 3 noise levels, seeds 100 + i;
    the n = 45 meshes (40 500 faces) are patch-split.  Raw meshes resident in HBM; weights = the bench's (random init + the
    timed steps), so the angles say nothing -- tools/test_synthetic.py reports them for a trained network."""
    from geobi_gnn_amd import meshgen, patches
    freqs, sigmas = (16, 22, 32, 45), (0.1, 0.2, 0.3)
    meshes = []
    for i in range(29):
        noisy, clean, faces = meshgen.noisy_icosphere(freqs[i % 4], sigmas[i % 3], seed=100 + i)
        meshes.append((torch.as_tensor(noisy, dtype=torch.float32, device=device),
                       torch.as_tensor(faces, dtype=torch.int32, device=device),
                       torch.as_tensor(clean, dtype=torch.float32, device=device)))
    edges = sum(16 * f.shape[0] + p.shape[0] - 60 for p, f, _ in meshes)       # E_v0 + E_f0 = (3F + V) + (13F - 60)

    def run(stats=None):
        split = 0
        for pts, fv, gt in meshes:
            r = patches.predict_mesh(net, pts, fv, sub_size=sub_size, n_iter=60, gt_points=gt, stats=stats)
            split += r['n_patches'] > 1
        return split
    run()                                                                    # warm-up (arenas, first launches)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        n_split = run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    stats = {}
    run(stats)                                                               # one more pass with a sync behind every phase
    tot = sum(stats.values())
    # the same list with several meshes in flight (patches.predict_many: one host thread + stream per worker)
    many = {}
    for workers in (2, 4):
        lst = [(p, f, g) for p, f, g in meshes]
        patches.predict_many(net, lst, workers=workers, sub_size=sub_size, n_iter=60)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        patches.predict_many(net, lst, workers=workers, sub_size=sub_size, n_iter=60)
        torch.cuda.synchronize()
        many['workers_%d' % workers] = round((time.perf_counter() - t1) * 1e3, 2)
    # the same list through patches.predict_batch: the small meshes (<= sub_size faces) as disjoint unions of up to 100 000
    # faces, the patch-split ones four at a time (growth chains side by side, passes of up to 12 pooled patches)
    patches.predict_batch(net, meshes, max_faces=100000, sub_size=sub_size, n_iter=60)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(reps):
        patches.predict_batch(net, meshes, max_faces=100000, sub_size=sub_size, n_iter=60)
    torch.cuda.synchronize()
    batched = (time.perf_counter() - t1) / reps
    log('test list with 2 / 4 meshes in flight: %s ms; batched (predict_batch): %.1f ms' % (many, batched * 1e3))
    log('test list: %.1f ms for 29 meshes (%d patch-split), %.1f M-edges/s' % (dt * 1e3, n_split, edges / dt / 1e6))
    return {'workload': 'Synthetic test_list stand-in: 29 noisy icospheres n in {16,22,32,45} x 3 noise levels through '
                        'patches.predict_mesh (preprocessing, patch split at %d faces, network, merge, 60-sweep vertex '
                        'update, angular errors), raw meshes resident in HBM' % sub_size,
            'total_ms': round(dt * 1e3, 2), 'meshes_per_s': round(29 / dt, 1), 'M_edges_per_s': round(edges / dt / 1e6, 1),
            'level0_edges': edges, 'faces': sum(f.shape[0] for _, f, _ in meshes), 'patch_split_meshes': n_split,
            'meshes_in_flight_total_ms': many,
            'batched': {'total_ms': round(batched * 1e3, 2), 'meshes_per_s': round(29 / batched, 1),
                        'M_edges_per_s': round(edges / batched / 1e6, 1),
                        'note': 'patches.predict_batch: the 15 meshes below the patch size through the network and the vertex '
                                'update as disjoint unions of <= 100 000 faces; the 14 patch-split ones four at a time (growth '
                                'chains side by side, network passes of <= 12 pooled patches, one vertex update per group); '
                                'per-mesh results bit-identical to the mesh-by-mesh run above'},
            'share_outside_network': round(1.0 - stats.get('network', 0.0) / tot, 3),
            'phase_ms_synchronised': {k: round(v * 1e3, 2) for k, v in stats.items()},
            'phase_note': 'one extra pass with a device sync behind every phase: upper bounds of the overlapped costs'}


def measure_groups(net, bucket, opt, parts_by_groups, steps=20, warm=5):
    """The timed step again with the batch's 4 meshes as mesh groups in flight together (executor.TrainGroups:
    geobi_net_train_groups), same box, right after the headline measurement: ms per step for 2 groups of 2 meshes."""
    from geobi_gnn_amd.executor import TrainGroups
    out = {}
    for g, parts in parts_by_groups.items():
        tg = TrainGroups(net, bucket, 'L1', 'L1').set_groups(parts)
        for _ in range(warm):
            grouped_step(tg, bucket, opt, collective=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            grouped_step(tg, bucket, opt, collective=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        out['groups_%d' % g] = {'ms_per_step': round(dt * 1e3, 3), 'sequential_fallbacks': tg.sequential_steps}
        log('mesh groups %d: %.3f ms/step' % (g, dt * 1e3))
    out['note'] = ('same batch, same step (forward, loss, backward, Adam; no collective), the meshes as G groups on G '
                   'streams / host threads, gradient buckets summed in group order; the headline value above is the '
                   'single-union step (--groups 1)')
    return out


def measure_collective(bucket, before=None, after=None, steps=5):
    """N > 1: the gradient all-reduce alone, in steps of their own after the timed region: `before()` refills the
    bucket (forward + backward), then bucket.all_reduce_mean() is bracketed by HIP events on the compute stream
    (wall clock for CPU tensors: the gloo plumbing test).  Median over `steps`, microseconds."""
    us = []
    cuda = bucket.flat.is_cuda
    for _ in range(steps):
        if before is not None:
            before()
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.all_reduce_mean()
            e1.record()
        else:
            t0 = time.perf_counter()
            bucket.all_reduce_mean()
            t1 = time.perf_counter()
        if after is not None:
            after()
        if cuda:
            torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3)
        else:
            us.append((t1 - t0) * 1e6)
    us.sort()
    return us[len(us) // 2]


def multi_rank_fields(my_ms, coll_us, device):
    """What a scaling line needs to explain itself (config.* at N > 1): backend, every rank's own step time (min /
    max over ranks) and the all-reduce alone (max over ranks)."""
    import torch.distributed as dist
    r = torch.tensor([my_ms], dtype=torch.float64, device=device)
    lo, hi = r.clone(), r.clone()
    c = torch.tensor([coll_us], dtype=torch.float64, device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.MAX)
    return {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
            'collective_us': round(float(c.item()), 1),
            'collective_note': 'median of 5 all-reduces of the flat gradient bucket incl. the division by the world '
                               'size (HIP events on the compute stream), max over ranks; separate steps after the '
                               'timed region',
            'rank_ms_per_step': {'min': round(float(lo.item()), 3), 'max': round(float(hi.item()), 3)}}


def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a 1-GPU box
    exposes every host CPU in the mask but grants a 16-core share; oversubscribing OpenMP stalls)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            pr = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // pr))
        except (OSError, ValueError):
            pass
    return min(n, 16)


def log(msg):
    print('[bench] ' + msg, file=sys.stderr, flush=True)


PMC_SUMMARY = 'profiles/r04_pmc_feast_fused.json'
PMC_MFMA = 'profiles/r04_pmc_mfma.json'


def pmc_traffic(kernel):
    """HBM bytes per launch from the PMC counters.  They cannot be collected from inside this
    process (rocprofv3 --pmc passes, FETCH_SIZE and WRITE_SIZE separately); the committed summary of
    those passes over this same workload is quoted, or null when it is absent."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    try:
        k = json.load(open(path))['kernels'][kernel]
        return k['hbm_bytes_per_launch'], 'quoted from %s (tools/pmc_summary.py)' % PMC_SUMMARY
    except (OSError, KeyError, ValueError):
        return None, None


def pmc_mfma():
    """MFMA counters per kernel family, QUOTED from the committed pass (counters cannot be read in-process): achieved
    TFLOP/s from SQ_INSTS_VALU_MFMA_MOPS_F32 over the dispatch time, fraction of the fp32 MFMA peak, MFMA-busy share."""
    try:
        ks = json.load(open(os.path.join(ROOT, PMC_MFMA)))['kernels']
    except (OSError, KeyError, ValueError):
        return None
    return {k.replace(' (all instantiations)', ''): {'tflops': v['tflops'], 'frac': v['frac_of_peak'],
                                                      'mfma_busy': v['mfma_busy'], 'avg_us': v['avg_us']}
            for k, v in ks.items() if k.endswith('(all instantiations)')}


def cpu_baseline(freq=FREQ, timed=3, warm=1):
    """PyG-shaped oracle, fp32, all host cores, ONE mesh of the bench size (bounded sample)."""
    from geobi_gnn_amd import meshgen
    from oracle import ref_model as R, pyg_ops as P
    cores = host_cores()
    torch.set_num_threads(cores)
    dv, df = meshgen.synthetic_dual_data(freq, 0.2, seed=200)
    edges = dv.edge_index.shape[1] + df.edge_index.shape[1]
    torch.manual_seed(0)
    net = R.DualGNN()
    times = []
    for it in range(warm + timed):
        a = P.Data(dv.x.clone(), dv.edge_index, edge_weight=dv.edge_weight, y=dv.y)
        b = P.Data(df.x.clone(), df.edge_index, edge_weight=df.edge_weight, y=df.y, fv_indices=df.fv_indices)
        net.zero_grad()
        t0 = time.perf_counter()
        vp, npred, _ = net((a, b))
        loss = R.dual_loss(R.loss_v(vp, a.y, 'L1'), R.loss_n(npred, b.y, 'L1'))
        loss.backward()
        dt = time.perf_counter() - t0
        log('cpu oracle pass %d: %.2f s on %d threads' % (it, dt, cores))
        if it >= warm:
            times.append(dt)
    med = sorted(times)[len(times) // 2]
    return {'value': round(edges / med / 1e6, 4), 'unit': 'M-edges/s', 'cores': cores, 'kind': 'port',
            'sample': '1 icosphere n=%d (F=%d, %d edges), fwd+loss+bwd, median of %d after %d warm-ups, %.2f s each; '
                      'torch %s CPU, PyG-shaped per-edge op decomposition (oracle/)' %
                      (freq, 20 * freq * freq, edges, timed, warm, med, torch.__version__)}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE): start the N ranks as a CHILD
    `python -m torch.distributed.run` process, relay rank 0's JSON line, return the child's exit code.

    Runs before anything in this process touches the GPU (torch.cuda.device_count() does not initialise
    it on this image) and never exec()s.  When the box shows fewer devices than ranks (the 1-GPU lease),
    the ranks share device 0 over gloo -- a REHEARSAL of the multi-rank path, labelled as such in the
    line; it is not a scaling measurement."""
    import subprocess
    n = args.gpus
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    have = torch.cuda.device_count()
    if have < n:
        if n > 4:
            log('only %d device(s) visible: a %d-rank rehearsal on one device exceeds the process guard of a 1-GPU box' % (have, n))
            return 2
        log('only %d device(s) visible for %d ranks: rehearsal on device 0 over gloo' % (have, n))
        env['GEOBI_ALL_RANKS_ON_DEVICE0'] = '1'
        env['GEOBI_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    log('launching: ' + ' '.join(cmd))
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:
        out = out.rstrip('\n')
        if out.startswith('{') and '"metric"' in out:
            line = out
        elif out:
            log('[child] ' + out)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        log('the ranks exited cleanly but printed no result line')
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--freq', type=int, default=FREQ, help=argparse.SUPPRESS)
    ap.add_argument('--groups', type=int, default=int(os.environ.get('GEOBI_BENCH_GROUPS', '1')),
                    help='mesh groups in flight together per step (1: the batch as one union graph on one stream)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip extra.infer / extra.fresh_batch')
    ap.add_argument('--plumbing', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch.distributed as dist
    from geobi_gnn_amd import network, _lib
    from geobi_gnn_amd.parallel import init_distributed, FlatParameters

    rank, world, device = init_distributed()
    if args.plumbing:
        # launcher / rendezvous / collective plumbing only (tests/test_host_logic.py drives the N > 1 branch on a
        # box without a GPU): no workload runs and no number is reported
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        cfg = {}
        if world > 1:
            from geobi_gnn_amd.parallel import GradBucket
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            # the N > 1 fields of the real line, through the same helpers, on a bucket of the real size (host memory)
            pb = GradBucket([torch.nn.Parameter(torch.zeros(939128))])
            pb.flat.fill_(float(rank + 1))
            t0 = time.perf_counter()
            us = measure_collective(pb, steps=3)
            cfg = multi_rank_fields((time.perf_counter() - t0) * 1e3 / 3, us, torch.device('cpu'))
            cfg['bucket_mean'] = float(pb.flat[0])
            dist.barrier()
        if rank == 0:
            print(json.dumps({'metric': 'M-edges/s (fwd+bwd) on Synthetic set', 'value': None, 'n_gpus': world,
                              'plumbing': True, 'rank_sum': t.item(), 'backend': dist.get_backend() if world > 1 else None,
                              'config': cfg}),
                  flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if world != args.gpus:
        raise SystemExit('bench.py: WORLD_SIZE=%d but --gpus %d (launch through torch.distributed.run '
                         '--nproc-per-node %d, or from a bare shell)' % (world, args.gpus, args.gpus))
    assert torch.cuda.is_available(), 'bench.py measures the MI355X path; no CPU fallback exists'
    rehearsal = os.environ.get('GEOBI_ALL_RANKS_ON_DEVICE0') == '1' and world > 1
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        # a scaling line must come from RCCL with one device per rank; gloo / shared device only as a labelled rehearsal
        assert rehearsal or (dist.get_backend() == 'nccl' and torch.cuda.device_count() >= world), \
            'N > 1 outside a rehearsal needs backend nccl (RCCL) and %d devices: backend %s, %d visible' % (
                world, dist.get_backend(), torch.cuda.device_count())
    _lib.lib()

    torch.manual_seed(0)                                  # random-init weights of the real architecture
    net = network.DualGNN().to(device)
    flat = FlatParameters(net)            # one flat parameter + one flat gradient bucket
    bucket = flat.bucket
    # same update rule as the reference's torch.optim.Adam(lr=1e-3) (train_dual.py:162), one launch over the flat
    # parameter (train_util.FlatAdam; parity with torch.optim.Adam: tests/test_gpu_model.py)
    from geobi_gnn_amd.train_util import FlatAdam
    opt = FlatAdam(flat.parameters(), lr=1e-3) if torch.cuda.is_available() else torch.optim.Adam(flat.parameters(), lr=1e-3)
    assert args.groups in (1, 2, 4), '--groups: 1, 2 or 4 (the batch has 4 meshes)'
    if world > 1 and os.environ.get('GEOBI_SPLIT_ALLREDUCE') == '1':
        enable_split_allreduce(net, bucket, device)
    dv, df, edges, parts = make_batch(rank, device, args.freq, groups=args.groups)
    tg = None
    if args.groups > 1:
        from geobi_gnn_amd.executor import TrainGroups
        tg = TrainGroups(net, bucket, 'L1', 'L1').set_groups(parts)

    def one_step():
        if tg is not None:
            return grouped_step(tg, bucket, opt)
        return train_step(net, bucket, opt, dv, df)

    log('rank %d: batch resident (%d edges), warming up' % (rank, edges))
    for _ in range(args.warmup):
        one_step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    e = torch.tensor([float(edges)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
    my_ms = elapsed / args.steps * 1e3
    elapsed, total_edges = float(t.item()), float(e.item())
    mr = None
    if world > 1:
        # the line must explain a scaling loss by itself: every rank's own step time and the all-reduce alone
        def refill():
            from geobi_gnn_amd.parallel import batched_losses
            bucket.zero()
            a, b = dv.shallow_copy(), df.shallow_copy()
            vp, npred, _ = net((a, b))
            lv, ln = batched_losses(vp, npred, dv, df, 'L1', 'L1')
            network.dual_loss(lv, ln).backward()
        mr = multi_rank_fields(my_ms, measure_collective(bucket, refill, opt.step), device)
    if rank == 0:
        log('timed %d steps: %.3f ms/step' % (args.steps, elapsed / args.steps * 1e3))

    out = {
        'metric': 'M-edges/s (fwd+bwd) on Synthetic set',
        'value': round(total_edges * args.steps / elapsed / 1e6, 2), 'unit': 'M-edges/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'Synthetic train, batch=4 meshes (~20k faces each), fwd+bwd '
                               '[BASELINE.json configs[2]]: per rank 4 noisy icospheres n=%d (F=%d) as one '
                               'disjoint-union graph, L1/L1 loss, grad all-reduce + Adam step inside the timed '
                               'region; ONE resident batch replayed every step, its level-0 CSR / reverse-edge index / '
                               'corner lists / loss weights built in warm-up and cached per mesh (extra.fresh_batch: '
                               'other meshes every step)' % (args.freq, 20 * args.freq ** 2),
                   'meshes_per_rank': BATCH, 'edges_per_rank_step': edges, 'parallelism': 'dp%d' % world,
                   'mesh_groups_in_flight': args.groups, 'split_allreduce': SPLIT_ALLREDUCE is not None,
                   'collective': ('none' if world == 1 else
                                  '%s all-reduce of the flat fp32 gradient bucket' % dist.get_backend()),
                   'final_loss': round(float(loss.sum().item()), 6)},
    }
    if world > 1:
        out['config'].update(mr)
    if rehearsal:
        out['config']['rehearsal'] = ('%d ranks share device 0 over gloo (fewer devices than ranks): checks the '
                                      'multi-rank path, NOT a scaling measurement' % world)
    if rank == 0:
        if not args.no_roofline:
            out['roofline'] = measure_roofline(net, bucket, opt, dv, df)
            out['roofline_mfma'] = measure_mfma(net, bucket, opt, dv, df)
        if world == 1 and not args.no_extra:
            out['extra'] = {'infer': measure_infer(net, device),
                            'fresh_batch': measure_fresh_batch(net, bucket, opt, device, args.freq),
                            'test_list': measure_test_list(net, device)}
            if args.groups == 1:
                _, _, _, parts2 = make_batch(rank, device, args.freq, groups=2)
                out['extra']['mesh_groups'] = measure_groups(net, bucket, opt, {2: parts2})
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.freq)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
