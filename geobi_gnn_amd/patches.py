"""Patch split / merge for meshes larger than one network pass.

* split  /root/reference/code/dataset.py:156-193 (``process_one_data``: seed = face farthest from the
         centroid, grow ``submesh_size`` faces ring by ring, next seed = farthest unvisited face) with
         /root/reference/code/data_util.py:55-84 ``mesh_get_neighbor_np`` and :318-336 ``get_submesh``
* merge  /root/reference/code/test_dual.py:49-61 (sum overlapping predictions, divide by the visit count,
         re-normalise the normals) followed by the 60-sweep vertex update (:63-72)

Everything runs on the device: the ordered ring growth (``geobi_patch_grow``: one workgroup per patch walks the rings,
the visiting order comes from slot numbers; the chain of patches runs on a side stream), vertex renumbering, graph
construction, the network and the merge.  The reference walks openmesh's ``vf_indices`` rows; here the incidence
lists are in ascending face order, so which faces the LAST, partial ring of a patch contributes can differ from an
openmesh run (whole rings do not depend on the order).
"""
import ctypes

import numpy as np
import torch

from . import _lib as L
from . import meshprep, network
from .data_util import computer_face_normal, update_position2
from .infer import predict_one_submesh


_GROW_STREAMS = {}


def _grow_stream(dev, slot=0):
    """One side stream per device for the growth chain: it runs beside the patches' own work on the caller's stream.
    slot: further chains of the same host thread (predict_batch grows the patches of several meshes at once)."""
    import threading
    key = (dev.type, dev.index, threading.get_ident(), slot)     # one chain per host thread (predict_many keeps its threads)
    if key not in _GROW_STREAMS:
        _GROW_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _GROW_STREAMS[key]


def patch_grow(fv, vf32, seed, neighbor_count=None, ring_count=None):
    """data_util.mesh_get_neighbor_np on the device (geobi_patch_grow), one patch from a given seed: fv [F,3] int32,
    vf32 [V,maxval] int32 padded incidence -> face ids (device int32) in the reference's visiting order."""
    dev = fv.device
    F, V = int(fv.shape[0]), int(vf32.shape[0])
    state = torch.empty(L.size_query('geobi_patch_grow_state_ints', F, V), dtype=torch.int32, device=dev)
    d2 = torch.zeros(F, dtype=torch.float32, device=dev)
    L.call('geobi_patch_grow_init', L.ptr(state), F, V, L.ptr(d2), L.stream())
    out = torch.empty(F, dtype=torch.int32, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    L.call('geobi_patch_grow', L.ptr(fv), L.ptr(vf32), int(vf32.shape[1]), F, V, None, int(seed), int(neighbor_count or 0),
           int(ring_count or 0), 1, L.ptr(state), L.ptr(out), L.ptr(n), None, 0, L.stream())
    return out[:L.read_i32(n, 1)[0]]


def _submesh_launch(fv, sel, num_vertices):
    """geobi_submesh enqueued: (V_idx buffer, F_sub, count [1]) -- the count is still on the device."""
    dev = fv.device
    n, V = int(sel.shape[0]), int(num_vertices)
    v_idx = torch.empty(min(V, 3 * n), dtype=torch.int32, device=dev)
    f_sub = torch.empty((n, 3), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = L.workspace(L.size_query('geobi_submesh_ws_bytes', n, V), dev)
    L.call('geobi_submesh', L.ptr(fv), L.ptr(sel), n, V, L.ptr(v_idx), L.ptr(f_sub), L.ptr(count), L.ptr(ws),
           ws.numel(), L.stream())
    return v_idx, f_sub, count


def submesh(fv, sel, num_vertices):
    """data_util.get_submesh on the device: sel [n] int32 face ids -> (V_idx [nv] int32, F_sub [n,3] int32)."""
    v_idx, f_sub, count = _submesh_launch(fv, sel, num_vertices)
    return v_idx[:L.read_i32(count, 1)[0]], f_sub


def build_patch_union(pts, fv, sels, centroid, scale, data_type='Synthetic'):
    """Several patches (face lists `sels`) of one mesh as ONE preprocessed disjoint-union pair: vertex renumbering of every
    patch (geobi_submesh) enqueued first and their sizes read TOGETHER, the patches' points and faces concatenated (two
    geobi_concat32 launches + one gather), then a single pass of meshprep.build_dual_data over the union -- graph
    construction and normals are local to a component; the bilateral weights take each patch's own mean edge length
    (geobi_calc_weight_parts), so every patch gets the bits it gets on its own.  What was ~35 small launches and two host
    reads PER PATCH is that per network pass.
    -> ((data_v, data_f), [v_idx per patch], vertex ranges, face ranges)"""
    from .data import _Concat
    dev, V = fv.device, int(pts.shape[0])
    subs = [_submesh_launch(fv, sel, V) for sel in sels]
    counts = L.read_i32(torch.cat([c for _, _, c in subs]))
    vptr, fptr = [0], [0]
    for nv, sel in zip(counts, sels):
        vptr.append(vptr[-1] + nv)
        fptr.append(fptr[-1] + int(sel.shape[0]))
    v_idx = [vi[:nv] for (vi, _, _), nv in zip(subs, counts)]
    if len(sels) == 1:
        idx_all, faces = v_idx[0], subs[0][1]
    else:
        cc = _Concat(dev)
        idx_all = cc.cat(v_idx)
        faces = cc.cat([f.view(-1) for _, f, _ in subs], vptr[:-1]).view(-1, 3)
        cc.run()
    dual = meshprep.build_dual_data(pts[idx_all.long()], faces, name='patches', data_type=data_type, device=dev,
                                    centroid=centroid, scale=scale, trusted_faces=True, want_vf=False,
                                    parts=(vptr, fptr) if len(sels) > 1 else None)
    return dual, v_idx, list(zip(vptr[:-1], vptr[1:])), list(zip(fptr[:-1], fptr[1:]))


def split_faces(points, fv, submesh_size, incidence=None, vf32=None, ahead=None):
    """The face lists of dataset.py:156-193's patches, grown on the device: generator of (seed-ordered) device int32
    tensors.  The whole split is a CHAIN of geobi_patch_grow launches on a side stream -- each grows one patch and picks the
    next seed (farthest unvisited face) -- enqueued `ahead` patches at a time; a patch's size reaches the host through
    mapped memory when its kernel ends, so the host only ever waits for the patch it is about to use, and the chain runs
    beside whatever the caller does with the earlier patches on its own stream."""
    dev = fv.device
    V, F = int(points.shape[0]), int(fv.shape[0])
    sub = int(min(submesh_size, F))
    if vf32 is None:
        rowptr, lst = incidence if incidence is not None else meshprep.vertex_faces(fv, V)
        vf32 = meshprep.vf_padded32(rowptr, lst, V)
    W = int(vf32.shape[1])
    centroid = points.mean(0, keepdim=True)
    d2 = ((points[fv.long()].mean(1) - centroid) ** 2).sum(1).contiguous()
    state = torch.empty(L.size_query('geobi_patch_grow_state_ints', F, V), dtype=torch.int32, device=dev)
    L.call('geobi_patch_grow_init', L.ptr(state), F, V, L.ptr(d2), L.stream())
    main, gs = torch.cuda.current_stream(dev), _grow_stream(dev)
    gs.wait_stream(main)
    ahead = int(ahead or min(32, 2 * (F + sub - 1) // sub + 2))
    k = 0
    try:
        while True:
            box_ptr = ctypes.c_void_p(0)
            L.call('geobi_host_mailbox', ahead, ctypes.byref(box_ptr))
            box = (ctypes.c_int32 * ahead).from_address(box_ptr.value)
            with torch.cuda.stream(gs):
                slab = torch.empty(ahead * sub, dtype=torch.int32, device=dev)
                events = []
                for i in range(ahead):
                    L.call('geobi_patch_grow', L.ptr(fv), L.ptr(vf32), W, F, V, L.ptr(d2), -1, sub, 0, k + i + 1, L.ptr(state),
                           slab.data_ptr() + 4 * i * sub, None, box_ptr.value + 4 * i, 1, gs.cuda_stream)
                    ev = torch.cuda.Event()
                    ev.record(gs)
                    events.append(ev)
            slab.record_stream(main)
            for i in range(ahead):
                # the wait runs inside the library (no interpreter lock held: other host threads keep working)
                word = L.lib().geobi_host_mailbox_wait(ctypes.c_void_p(box_ptr.value + 4 * i), ctypes.c_void_p(gs.cuda_stream))
                if word == 0:
                    raise L.GeobiError('patch growth: the chain ended without reporting patch %d' % (k + i))
                n = (word - 1) // 2
                if n == 0:
                    return
                main.wait_event(events[i])
                yield slab[i * sub:i * sub + n]
            k += ahead
    finally:
        # whatever ends the walk (the last patch, an error, a consumer that stops early): nothing of the chain may still be
        # running when its buffers and the mailbox go back
        gs.synchronize()


def split_patches(points, fv, submesh_size, incidence=None, keep=None, vf32=None):
    """Generator over the patches of dataset.py:156-193: yields (select_faces, V_idx, F_sub) device int32
    tensors.  points [V,3] fp32 and fv [F,3] int32 live on the device; the face lists come from the device growth chain
    (split_faces), vertex renumbering (get_submesh, data_util.py:318-336) from geobi_submesh.
    keep: optional callable(k) -> bool.  Patch k is still grown (the next seed depends on every earlier patch) but
    when keep(k) is false nothing is built for it and None is yielded -- how the ranks of a multi-GPU inference each
    take their share of one mesh's patches."""
    V = points.shape[0]
    for k, sel in enumerate(split_faces(points, fv, submesh_size, incidence=incidence, vf32=vf32)):
        if keep is None or keep(k):
            v_idx, f_sub = submesh(fv, sel, V)
            yield sel, v_idx, f_sub
        else:
            yield None


def _union_dual(duals):
    """Several device-built (data_v, data_f) pairs -> one pair over the disjoint union of their graphs (CSR
    concatenated directly); returns the pair and the per-patch (vertex, face) row ranges."""
    from .data import union_batch_graphs
    data_v, data_f = union_batch_graphs(duals)
    pv, pf = data_v.mesh_ptr.tolist(), data_f.mesh_ptr.tolist()
    return (data_v, data_f), list(zip(pv[:-1], pv[1:])), list(zip(pf[:-1], pf[1:]))


class _Phases(object):
    """Optional phase clock of predict_mesh (``stats`` dict): wall time per phase with a device sync behind each, so
    the shares are upper bounds of what the phases cost when they overlap.  Without ``stats`` nothing is synchronised."""

    def __init__(self, stats):
        self.stats = stats
        if stats is not None:
            import time
            self.clock = time.perf_counter
            torch.cuda.synchronize()
            self.t = self.clock()

    def tick(self, name):
        if self.stats is not None:
            torch.cuda.synchronize()
            now = self.clock()
            self.stats[name] = self.stats.get(name, 0.0) + (now - self.t)
            self.t = now


def predict_mesh(net, points, faces, sub_size=20000, n_iter=60, data_type='Synthetic', gt_points=None, patch_batch=5,
                 distributed=None, stats=None):
    """test_dual.py:24-87 without the OBJ IO, for a mesh of any size: preprocessing, patch split when
    F > sub_size, network, merge, de-normalisation, vertex update -- all device-resident.  The reference runs
    the patches one by one; here `patch_batch` of them go through the network as one disjoint-union graph
    (independent components: same results, fewer and better-filled launches).

    Multi-GPU: with torch.distributed initialised (``distributed=None`` picks that up; False turns it
    off) every rank calls this with the same mesh; the patches are dealt round-robin over the ranks
    (parallel.owns_patch), each rank merges its own into its Vp / Np / visit-count sums, ONE reduction to rank 0
    (parallel.reduce_patch_sums) adds them, and rank 0 finalises and runs the vertex update.  No graph is split
    across GPUs and nothing but those three sums crosses xGMI.  Other ranks get Vp = Np = V_updated = None.

    stats: optional dict; seconds per phase are ADDED to its keys 'preprocessing' (incidence, graphs, weights,
    features -- of the mesh or of its patches), 'growth' (patch split: ring growth, seeds, vertex renumbering),
    'network', 'merge+update' (overlap merge, de-normalisation, vertex update, errors).

    Returns dict(Vp, Np, V_updated, n_patches, angle1, angle2)."""
    from . import parallel
    rank, world = parallel.rank_world() if distributed is None or distributed else (0, 1)
    dev = next(net.parameters()).device
    pts = torch.as_tensor(np.asarray(points) if not torch.is_tensor(points) else points)
    pts = pts.to(device=dev, dtype=torch.float32).contiguous()
    fv = torch.as_tensor(np.asarray(faces) if not torch.is_tensor(faces) else faces).to(device=dev, dtype=torch.int32)
    fv = fv.contiguous()
    V, F = pts.shape[0], fv.shape[0]
    ph = _Phases(stats)
    if F > 0:                          # the caller's face table: range-checked before any kernel walks it
        lo, hi = L.read_i32(torch.cat([t.reshape(1) for t in torch.aminmax(fv)]))
        if lo < 0 or hi >= V:
            raise L.GeobiError('faces index vertices outside [0, %d)' % V)
    vf = None
    if F <= sub_size:
        n_patches = 1
        if rank != 0:                  # one patch: nothing to share out
            return {'Vp': None, 'Np': None, 'V_updated': None, 'n_patches': 1, 'angle1': None, 'angle2': None}
        # the mesh's own preprocessing yields incidence, centroid, scale and the padded vf table on its way
        dual = meshprep.build_dual_data(pts, fv, name='mesh', data_type=data_type, device=dev, trusted_faces=True)
        meta = dual[0].meta
        centroid, scale, vf = meta['centroid'], meta['scale'], meta['vf_indices']
        rowptr, lst = meta['incidence']
        ph.tick('preprocessing')
        Vp, Np = predict_one_submesh(net, dual)
        ph.tick('network')
        Vp = Vp / scale + centroid
    else:
        rowptr, lst = meshprep.vertex_faces(fv, V)
        vf32 = meshprep.vf_padded32(rowptr, lst, V)
        vf = vf32.long()
        g_v = meshprep.ring_graph(0, fv, rowptr, lst, V)
        centroid = pts.mean(0, keepdim=True)
        host = torch.cat([meshprep.mean_edge_length(pts, g_v), centroid.reshape(-1)]).tolist()     # one read for both
        scale, c = float(1.0 / torch.tensor(host[0], dtype=torch.float32)), host[1:]
        ph.tick('preprocessing')
        Vp = torch.zeros((V, 3), dtype=torch.float32, device=dev)
        Np = torch.zeros((F, 3), dtype=torch.float32, device=dev)
        sum_v = torch.zeros(V, dtype=torch.int32, device=dev)
        n_patches = 0
        pending = []

        def flush():
            if not pending:
                return
            dual, v_idxs, vr, fr = build_patch_union(pts, fv, pending, centroid, scale, data_type)
            ph.tick('preprocessing')
            vert_p, norm_p = predict_one_submesh(net, dual)
            ph.tick('network')
            for sel, v_idx, (v0, v1), (f0, f1) in zip(pending, v_idxs, vr, fr):
                L.call('geobi_patch_accumulate', L.ptr(vert_p[v0:v1]), L.ptr(norm_p[f0:f1]), L.ptr(v_idx), L.ptr(sel),
                       v_idx.shape[0], sel.shape[0], L.ptr(Vp), L.ptr(Np), L.ptr(sum_v), L.stream())
            del pending[:]
            ph.tick('merge+update')

        for k, sel in enumerate(split_faces(pts, fv, sub_size, incidence=(rowptr, lst), vf32=vf32)):
            n_patches += 1
            ph.tick('growth')
            if world > 1 and not parallel.owns_patch(k, rank, world):
                continue
            pending.append(sel)
            if len(pending) >= max(1, int(patch_batch)):
                flush()
        flush()
        if world > 1:
            parallel.reduce_patch_sums([Vp, Np, sum_v], dst=0)
            if rank != 0:
                return {'Vp': None, 'Np': None, 'V_updated': None, 'n_patches': n_patches, 'angle1': None,
                        'angle2': None}
        L.call('geobi_patch_finalize', L.ptr(Vp), L.ptr(Np), L.ptr(sum_v), V, F, scale, c[0], c[1], c[2], L.stream())

    dd = torch.nn.functional.normalize(pts, dim=1) if data_type in ('Kinect_v1', 'Kinect_v2') else None
    if vf is None:
        vf = meshprep.vf_padded(rowptr, lst, V)
    Vu = update_position2(Vp, fv, vf, Np, n_iter=n_iter, depth_direction=dd)
    out = {'Vp': Vp, 'Np': Np, 'V_updated': Vu, 'n_patches': n_patches, 'angle1': None, 'angle2': None}
    if gt_points is not None:
        gt = torch.as_tensor(np.asarray(gt_points) if not torch.is_tensor(gt_points) else gt_points)
        gt = gt.to(device=dev, dtype=torch.float32).contiguous()
        Nt = computer_face_normal(gt, fv)
        out['angle1'] = float(network.error_n(Np, Nt))
        out['angle2'] = float(network.error_n(computer_face_normal(Vu, fv), Nt))
    ph.tick('merge+update')
    return out


class _SplitJob(object):
    """One mesh above the patch size inside predict_batch: its whole-mesh preprocessing, its growth chain (ALL launches of
    the chain enqueued at once on a stream of its own, sizes through its share of the thread's mailbox) and the sums its
    patches are merged into -- predict_mesh's split branch cut into steps, so that several meshes can take them together."""

    def __init__(self, index, pts, fv, gt):
        self.index, self.pts, self.fv, self.gt = index, pts, fv, gt
        self.V, self.F = int(pts.shape[0]), int(fv.shape[0])

    def prepare(self):
        rowptr, lst = meshprep.vertex_faces(self.fv, self.V)
        self.incidence = (rowptr, lst)
        self.vf32 = meshprep.vf_padded32(rowptr, lst, self.V)
        g_v = meshprep.ring_graph(0, self.fv, rowptr, lst, self.V)
        self.centroid = self.pts.mean(0, keepdim=True)
        self.host_words = torch.cat([meshprep.mean_edge_length(self.pts, g_v), self.centroid.reshape(-1)])

    def normalisation(self, host):
        self.scale, self.c = float(1.0 / torch.tensor(host[0], dtype=torch.float32)), host[1:]

    def start_chain(self, sub_size, box_ptr, ahead, slot):
        dev = self.fv.device
        self.sub = int(min(sub_size, self.F))
        self.ahead, self.box_ptr = ahead, box_ptr
        d2 = ((self.pts[self.fv.long()].mean(1) - self.centroid) ** 2).sum(1).contiguous()
        state = torch.empty(L.size_query('geobi_patch_grow_state_ints', self.F, self.V), dtype=torch.int32, device=dev)
        L.call('geobi_patch_grow_init', L.ptr(state), self.F, self.V, L.ptr(d2), L.stream())
        main, gs = torch.cuda.current_stream(dev), _grow_stream(dev, slot)
        gs.wait_stream(main)
        self.gs, self.events = gs, []
        with torch.cuda.stream(gs):
            self.slab = torch.empty(ahead * self.sub, dtype=torch.int32, device=dev)
            for i in range(ahead):
                L.call('geobi_patch_grow', L.ptr(self.fv), L.ptr(self.vf32), int(self.vf32.shape[1]), self.F, self.V, L.ptr(d2),
                       -1, self.sub, 0, i + 1, L.ptr(state), self.slab.data_ptr() + 4 * i * self.sub, None, box_ptr + 4 * i, 1,
                       gs.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(gs)
                self.events.append(ev)
        self.slab.record_stream(main)
        self._keep = (d2, state)

    def collect(self):
        """-> the patches' face lists (waits for the chain), or None when the chain needs more launches than were enqueued
        (the caller then takes the mesh through predict_mesh)."""
        main = torch.cuda.current_stream(self.fv.device)
        sels = []
        for i in range(self.ahead):
            word = L.lib().geobi_host_mailbox_wait(ctypes.c_void_p(self.box_ptr + 4 * i), ctypes.c_void_p(self.gs.cuda_stream))
            if word == 0:
                raise L.GeobiError('patch growth: the chain ended without reporting patch %d' % i)
            n = (word - 1) // 2
            if n == 0:
                return sels
            main.wait_event(self.events[i])
            sels.append(self.slab[i * self.sub:i * self.sub + n])
        return None

    def sums(self):
        dev = self.fv.device
        self.Vp = torch.zeros((self.V, 3), dtype=torch.float32, device=dev)
        self.Np = torch.zeros((self.F, 3), dtype=torch.float32, device=dev)
        self.sum_v = torch.zeros(self.V, dtype=torch.int32, device=dev)


def _predict_split_group(net, jobs, results, sub_size, n_iter, data_type, patch_batch):
    """predict_mesh's patch-split branch for SEVERAL meshes at once (patches.predict_batch): the meshes' growth chains run
    beside one another (a chain is one workgroup per launch), their normalisation constants come back in one read, a network
    pass takes patches of any of them (per mesh one build_patch_union, then the union of those), and the vertex update runs
    once over the union of the meshes.  Every mesh gets the bits predict_mesh gives it alone."""
    dev = jobs[0].fv.device
    for j in jobs:
        j.prepare()
    host = torch.cat([j.host_words for j in jobs]).tolist()              # one read for all of them
    aheads = [int(min(32, 2 * ((j.F + sub_size - 1) // sub_size) + 2)) for j in jobs]
    box = ctypes.c_void_p(0)
    L.call('geobi_host_mailbox', sum(aheads), ctypes.byref(box))
    off = 0
    try:
        for slot, (j, a) in enumerate(zip(jobs, aheads)):
            j.normalisation(host[4 * slot:4 * slot + 4])
            j.start_chain(sub_size, box.value + 4 * off, a, slot)
            off += a
        live, pending, general = [], [], []
        collected = [(j, j.collect()) for j in jobs]                     # every chain's words are read before the mailbox is reused
        for j, sels in collected:
            if sels is None:                                             # more patches than launches: the general path, below
                general.append(j)
                continue
            j.sums()
            j.n_patches = len(sels)
            live.append(j)
            pending += [(j, sel) for sel in sels]
        # ---- network passes over the pooled patches, in order
        step = max(1, int(patch_batch))
        for p0 in range(0, len(pending), step):
            chunk = pending[p0:p0 + step]
            owners = []
            for j, _ in chunk:
                if j not in owners:
                    owners.append(j)
            parts = []
            for j in owners:
                sels = [sel for jj, sel in chunk if jj is j]
                dual, v_idxs, vr, fr = build_patch_union(j.pts, j.fv, sels, j.centroid, j.scale, data_type)
                parts.append((j, sels, dual, v_idxs, vr, fr))
            if len(parts) == 1:
                dual, voff, foff = parts[0][2], [0], [0]
            else:
                dual, vrs, frs = _union_dual([pt[2] for pt in parts])
                voff, foff = [v0 for v0, _ in vrs], [f0 for f0, _ in frs]
            vert_p, norm_p = predict_one_submesh(net, dual)
            for (j, sels, _, v_idxs, vr, fr), vo, fo in zip(parts, voff, foff):
                for sel, v_idx, (v0, v1), (f0, f1) in zip(sels, v_idxs, vr, fr):
                    L.call('geobi_patch_accumulate', L.ptr(vert_p[vo + v0:vo + v1]), L.ptr(norm_p[fo + f0:fo + f1]), L.ptr(v_idx),
                           L.ptr(sel), v_idx.shape[0], sel.shape[0], L.ptr(j.Vp), L.ptr(j.Np), L.ptr(j.sum_v), L.stream())
        for j in live:
            L.call('geobi_patch_finalize', L.ptr(j.Vp), L.ptr(j.Np), L.ptr(j.sum_v), j.V, j.F, j.scale, j.c[0], j.c[1], j.c[2],
                   L.stream())
        for j in general:
            j.gs.synchronize()
            kw = {} if j.gt is None else {'gt_points': j.gt}
            results[j.index] = predict_mesh(net, j.pts, j.fv, sub_size=sub_size, n_iter=n_iter, data_type=data_type,
                                            patch_batch=patch_batch, distributed=False, **kw)
        if not live:
            return
        # ---- one vertex update over the union of the meshes (each vertex walks its own faces: same bits as per mesh)
        if len(live) == 1:
            j = live[0]
            Vp_u, Np_u, fv_u, vf_u = j.Vp, j.Np, j.fv, j.vf32
            vptr, fptr = [0, j.V], [0, j.F]
        else:
            vptr, fptr = [0], [0]
            for j in live:
                vptr.append(vptr[-1] + j.V); fptr.append(fptr[-1] + j.F)
            W = max(int(j.vf32.shape[1]) for j in live)
            Vp_u, Np_u = torch.cat([j.Vp for j in live]), torch.cat([j.Np for j in live])
            fv_u = torch.cat([j.fv + v0 for j, v0 in zip(live, vptr)])
            vf_u = torch.full((vptr[-1], W), -1, dtype=torch.int32, device=dev)
            for j, v0, f0 in zip(live, vptr, fptr):
                w = int(j.vf32.shape[1])
                vf_u[v0:v0 + j.V, :w] = torch.where(j.vf32 >= 0, j.vf32 + f0, j.vf32)
        dd = None
        if data_type in ('Kinect_v1', 'Kinect_v2'):
            dd = torch.nn.functional.normalize(torch.cat([j.pts for j in live]), dim=1)
        Vu = update_position2(Vp_u, fv_u, vf_u, Np_u, n_iter=n_iter, depth_direction=dd)
        for j, v0, f0 in zip(live, vptr, fptr):
            out = {'Vp': j.Vp, 'Np': j.Np, 'V_updated': Vu[v0:v0 + j.V], 'n_patches': j.n_patches, 'angle1': None, 'angle2': None}
            if j.gt is not None:
                g_ = torch.as_tensor(np.asarray(j.gt) if not torch.is_tensor(j.gt) else j.gt).to(device=dev, dtype=torch.float32).contiguous()
                Nt = computer_face_normal(g_, j.fv)
                out['angle1'] = network.error_n(j.Np, Nt)                 # device scalars: predict_batch reads them all at once
                out['angle2'] = network.error_n(computer_face_normal(out['V_updated'], j.fv), Nt)
            results[j.index] = out
    finally:
        for j in jobs:                        # nothing of a chain may still be running when its buffers and the mailbox go back
            if getattr(j, 'gs', None) is not None:
                j.gs.synchronize()


def predict_batch(net, meshes, max_faces=100000, sub_size=20000, n_iter=60, data_type='Synthetic', patch_batch=12,
                  split_group=4):
    """test_dual.py:90-148 (predict_dir) over a list of meshes with the SMALL ones -- at most ``sub_size`` faces, one
    network pass each in the reference -- going through the network and the vertex update several at a time, as one
    disjoint-union mesh of up to ``max_faces`` faces (in list order).  A single n = 32 mesh is a chain of
    ~110 launches and four size reads that takes 1.0 ms whatever its size (a 7 x larger mesh takes 2.1 ms): the union
    fills those launches.  Components do not interact -- graph construction and the weights stay per mesh, the network's
    rows are independent per component (tests/test_gpu_model.py: test_full_size_properties), the vertex update walks each
    vertex's own faces -- so every mesh gets the bits ``predict_mesh`` gives it alone (tests/test_gpu_patches.py).  Meshes
    above ``sub_size`` take predict_mesh's patch-split steps ``split_group`` at a time (_predict_split_group: growth chains
    side by side, network passes over the pooled patches ``patch_batch`` at a time, one vertex update); split_group = 1:
    predict_mesh one by one.  The 29-mesh stand-in list (tools/test_list_probe.py, ms for the list): mesh by mesh 88.7;
    small meshes as unions 73.0; + split groups of 2 / 5 patches per pass 64.9; groups of 4 / 12 patches per pass 46.8.
    meshes: list of (points, faces) or (points, faces, gt_points) -> list of predict_mesh's result dicts, in order."""
    dev = next(net.parameters()).device
    results = [None] * len(meshes)
    group, faces_in_group = [], 0
    big = []

    def flush_big():
        if big:
            _predict_split_group(net, list(big), results, sub_size, n_iter, data_type, patch_batch)
            del big[:]

    def flush():
        if not group:
            return
        duals = [g[3] for g in group]
        if len(group) == 1:
            dual, vr, fr = duals[0], [(0, group[0][1].shape[0])], [(0, group[0][2].shape[0])]
        else:
            dual, vr, fr = _union_dual(duals)
        Vp, Np = predict_one_submesh(net, dual)      # owned copies of the arena's results
        for (i, pts, fv, d, gt), (v0, v1) in zip(group, vr):
            meta = d[0].meta
            Vp[v0:v1] = Vp[v0:v1] / meta['scale'] + meta['centroid']
        if len(group) == 1:
            fv_u, vf_u, pts_u = group[0][2], group[0][3][0].meta['vf_indices'], group[0][1]
        else:
            from .data import _Concat
            cc = _Concat(dev)
            fv_u = cc.cat([g[2].view(-1) for g in group], [v0 for v0, _ in vr]).view(-1, 3)
            cc.run()
            rowptr, lst = meshprep.vertex_faces(fv_u, Vp.shape[0])
            vf_u = meshprep.vf_padded32(rowptr, lst, Vp.shape[0])
            pts_u = None
        dd = None
        if data_type in ('Kinect_v1', 'Kinect_v2'):
            dd = torch.nn.functional.normalize(torch.cat([g[1] for g in group]) if pts_u is None else pts_u, dim=1)
        Vu = update_position2(Vp, fv_u, vf_u, Np, n_iter=n_iter, depth_direction=dd)
        for (i, pts, fv, d, gt), (v0, v1), (f0, f1) in zip(group, vr, fr):
            out = {'Vp': Vp[v0:v1], 'Np': Np[f0:f1], 'V_updated': Vu[v0:v1], 'n_patches': 1, 'angle1': None, 'angle2': None}
            if gt is not None:
                g_ = torch.as_tensor(np.asarray(gt) if not torch.is_tensor(gt) else gt).to(device=dev, dtype=torch.float32).contiguous()
                Nt = computer_face_normal(g_, fv)
                out['angle1'] = network.error_n(out['Np'], Nt)            # device scalars, read all at once at the end
                out['angle2'] = network.error_n(computer_face_normal(out['V_updated'], fv), Nt)
            results[i] = out
        del group[:]

    for i, m in enumerate(meshes):
        gt = m[2] if len(m) > 2 else None
        pts = torch.as_tensor(np.asarray(m[0]) if not torch.is_tensor(m[0]) else m[0]).to(device=dev, dtype=torch.float32).contiguous()
        fv = torch.as_tensor(np.asarray(m[1]) if not torch.is_tensor(m[1]) else m[1]).to(device=dev, dtype=torch.int32).contiguous()
        F = fv.shape[0]
        if F > sub_size or F == 0:              # the group of small meshes stays open: results are placed by index
            if F == 0 or int(split_group) <= 1:
                kw = {} if gt is None else {'gt_points': gt}
                results[i] = predict_mesh(net, pts, fv, sub_size=sub_size, n_iter=n_iter, data_type=data_type,
                                          patch_batch=patch_batch, distributed=False, **kw)
                continue
            lo, hi = L.read_i32(torch.cat([t.reshape(1) for t in torch.aminmax(fv)]))     # range-checked before any kernel
            if lo < 0 or hi >= pts.shape[0]:
                raise L.GeobiError('faces index vertices outside [0, %d)' % pts.shape[0])
            big.append(_SplitJob(i, pts, fv, gt))
            if len(big) >= int(split_group):
                flush_big()
            continue
        if group and faces_in_group + F > max_faces:
            flush()
            faces_in_group = 0
        # the caller's face table is range-checked inside build_dual_data before any kernel walks it
        dual = meshprep.build_dual_data(pts, fv, name='mesh', data_type=data_type, device=dev)
        group.append((i, pts, fv, dual, gt))
        faces_in_group += F
    flush()
    flush_big()
    # the angular errors of all meshes in ONE host read (predict_mesh reads two scalars per mesh)
    slots = [(r, k) for r in results if r is not None for k in ('angle1', 'angle2') if torch.is_tensor(r[k])]
    if slots:
        vals = torch.stack([r[k].reshape(()) for r, k in slots]).tolist()
        for (r, k), v in zip(slots, vals):
            r[k] = v
    return results


_POOLS = {}          # workers -> ThreadPoolExecutor: the worker threads (= library contexts: mailbox, scan state, side streams,
_TLS = None          # growth stream) live as long as the process, so repeated calls allocate nothing new


def predict_many(net, meshes, workers=2, **kwargs):
    """test_dual.py:90-148 (predict_dir) over a list of meshes, `workers` of them in flight together: every worker is a
    host thread with its own stream, i.e. its own context of the library (arena, side streams, growth chain, mailbox are
    per host thread / per stream), so one mesh's host waits -- size reads, patch sizes -- sit under another mesh's kernels.
    The threads are kept (one pool per worker count).  meshes: list of (points, faces) or (points, faces, gt_points);
    kwargs as predict_mesh.  -> list of its results, in order."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    global _TLS
    if _TLS is None:
        _TLS = threading.local()
    dev = next(net.parameters()).device
    workers = max(1, min(int(workers), len(meshes)))
    if workers == 1:
        out = []
        for m in meshes:
            kw = dict(kwargs)
            if len(m) > 2 and m[2] is not None:
                kw['gt_points'] = m[2]
            out.append(predict_mesh(net, m[0], m[1], **kw))
        return out
    pool = _POOLS.get(workers)
    if pool is None:
        pool = _POOLS[workers] = ThreadPoolExecutor(max_workers=workers, thread_name_prefix='geobi-infer')
    main = torch.cuda.current_stream(dev)
    ready = torch.cuda.Event()
    ready.record(main)
    results = [None] * len(meshes)
    nxt = [0]
    lock = threading.Lock()

    def run():
        stream = getattr(_TLS, 'stream', None)
        if stream is None or stream.device != dev:
            stream = _TLS.stream = torch.cuda.Stream(device=dev)
        stream.wait_event(ready)
        with torch.cuda.stream(stream):
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= len(meshes):
                    break
                m = meshes[i]
                kw = dict(kwargs)
                if len(m) > 2 and m[2] is not None:
                    kw['gt_points'] = m[2]
                results[i] = predict_mesh(net, m[0], m[1], **kw)
        stream.synchronize()
    futures = [pool.submit(run) for _ in range(workers)]
    errors = [f.exception() for f in futures]
    for e in errors:
        if e is not None:
            raise e
    return results
