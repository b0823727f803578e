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


def make_batch(rank, device, freq=FREQ, batch=BATCH):
    from geobi_gnn_amd import meshgen
    from geobi_gnn_amd.data import union_batch
    pairs = []
    for i in range(batch):
        sigma = (0.1, 0.2, 0.3)[i % 3]                                   # the _n1/_n2/_n3 noise levels
        pairs.append(meshgen.synthetic_dual_data(freq, sigma, seed=200 + rank * batch + i))
    edges = sum(p[0].edge_index.shape[1] + p[1].edge_index.shape[1] for p in pairs)
    dv, df = union_batch(pairs)
    return dv.to(device), df.to(device), edges


def train_step(net, bucket, opt, dv0, df0, collective=True):
    from geobi_gnn_amd import network
    from geobi_gnn_amd.parallel import batched_losses
    bucket.zero()
    dv, df = dv0.shallow_copy(), df0.shallow_copy()      # the forward rewrites .x on the bag it gets
    vp, npred, _ = net((dv, df))
    lv, ln = batched_losses(vp, npred, dv0, df0, 'L1', 'L1')
    loss = network.dual_loss(lv, ln)
    loss.backward()
    if collective:
        bucket.all_reduce_mean()
    opt.step()
    return loss


def measure_roofline(net, bucket, opt, dv, df, steps=3):
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
    kname = 'feast_fused_kernel<%d,0,%d,%d>' % (best['cin'], lc, best['cout'] // 32)     # <C, MODE, LC, NT>
    avg_us = best['ms'] * 1e3 / best['launches']
    traffic, traffic_src = pmc_traffic(kname)
    # the same kernel against its other ceiling: the node transform on the fp32 matrix cores (2 N 9 Cin Cout flops per
    # launch; the dominant layer runs on both level-0 graphs, N averaged over its launches)
    n_avg = (dv.x.shape[0] + df.x.shape[0]) / 2.0
    tflops = 2.0 * n_avg * 9 * best['cin'] * best['cout'] / (avg_us * 1e-6) / 1e12
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
        'launches': best['launches'], 'avg_us': round(avg_us, 2),
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
           'pmc': 'profiles/r01_pmc_gemm_mfma.json (round-1 counters of these kernels: SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F32; tools/pmc_mfma.py)'}
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


PMC_SUMMARY = 'profiles/r02_pmc_feast_fused.json'


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


def cpu_baseline(freq=FREQ, timed=5, warm=2):
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
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
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
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.barrier()
        if rank == 0:
            print(json.dumps({'metric': 'M-edges/s (fwd+bwd) on Synthetic set', 'value': None, 'n_gpus': world,
                              'plumbing': True, 'rank_sum': t.item(), 'backend': dist.get_backend() if world > 1 else None}),
                  flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if world != args.gpus:
        raise SystemExit('bench.py: WORLD_SIZE=%d but --gpus %d (launch through torch.distributed.run '
                         '--nproc-per-node %d, or from a bare shell)' % (world, args.gpus, args.gpus))
    assert torch.cuda.is_available(), 'bench.py measures the MI355X path; no CPU fallback exists'
    if world > 1:
        assert dist.get_world_size() == args.gpus
    rehearsal = os.environ.get('GEOBI_ALL_RANKS_ON_DEVICE0') == '1' and world > 1
    _lib.lib()

    torch.manual_seed(0)                                  # random-init weights of the real architecture
    net = network.DualGNN().to(device)
    flat = FlatParameters(net)            # one flat parameter + one flat gradient bucket
    bucket = flat.bucket
    # same update rule as the reference's torch.optim.Adam(lr=1e-3) (train_dual.py:162), single-kernel form
    opt = torch.optim.Adam(flat.parameters(), lr=1e-3, fused=torch.cuda.is_available())
    dv, df, edges = make_batch(rank, device, args.freq)

    log('rank %d: batch resident (%d edges), warming up' % (rank, edges))
    for _ in range(args.warmup):
        train_step(net, bucket, opt, dv, df)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(net, bucket, opt, dv, df)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    e = torch.tensor([float(edges)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
    elapsed, total_edges = float(t.item()), float(e.item())
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
                               'region' % (args.freq, 20 * args.freq ** 2),
                   'meshes_per_rank': BATCH, 'edges_per_rank_step': edges, 'parallelism': 'dp%d' % world,
                   'collective': ('none' if world == 1 else
                                  '%s all-reduce of the flat fp32 gradient bucket' % dist.get_backend()),
                   'final_loss': round(float(loss.item()), 6)},
    }
    if rehearsal:
        out['config']['rehearsal'] = ('%d ranks share device 0 over gloo (fewer devices than ranks): checks the '
                                      'multi-rank path, NOT a scaling measurement' % world)
    if rank == 0:
        if not args.no_roofline:
            out['roofline'] = measure_roofline(net, bucket, opt, dv, df)
            out['roofline_mfma'] = measure_mfma(net, bucket, opt, dv, df)
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
