"""Binding of the whole-network forward (geobi_net_forward, include/geobi_hip.h): DualGNN's inference pass as one
library call.  The module-by-module path of network.py stays the reference for it (same kernels, same order,
bit-identical outputs) and takes over whenever a case is outside the fast path.

GEOBI_NET_EXECUTOR=0 turns it off (A/B timing; the parity tests run both)."""
import ctypes
import os

import torch

from . import _lib as L

ENABLED = os.environ.get('GEOBI_NET_EXECUTOR', '1') == '1'
STATS = {'calls': 0, 'fallback': 0, 'arena_retry': 0}      # how often the fast path was taken / left

_F = ctypes.c_void_p


class _Conv(ctypes.Structure):
    _fields_ = [('lin_w', _F), ('u_w', _F), ('c', _F), ('bias', _F)]


class _Gnn(ctypes.Structure):
    _fields_ = [('conv', _Conv * 8)]


class _Params(ctypes.Structure):
    _fields_ = [('gnn_v', _Gnn), ('gnn_f', _Gnn),
                ('fc_v1_w', _F), ('fc_v1_b', _F), ('fc_v2_w', _F), ('fc_v2_b', _F),
                ('fc_f1_w', _F), ('fc_f1_b', _F), ('fc_f2_w', _F), ('fc_f2_b', _F),
                ('force_depth', ctypes.c_int32), ('pool_mean', ctypes.c_int32)]


class _Level0(ctypes.Structure):
    _fields_ = [('N', ctypes.c_int64), ('E', ctypes.c_int64), ('rowptr', _F), ('col', _F), ('row', _F), ('weight', _F)]


class _Branch(ctypes.Structure):
    _fields_ = [('nodes', ctypes.c_int64 * 3), ('unpool_off', ctypes.c_int64 * 2),
                ('cluster_off', (ctypes.c_int64 * 2) * 2), ('cluster_len', (ctypes.c_int64 * 2) * 2)]


class _Out(ctypes.Structure):
    _fields_ = [('verts_off', ctypes.c_int64), ('normals_off', ctypes.c_int64), ('xf_off', ctypes.c_int64),
                ('used_bytes', ctypes.c_int64), ('v', _Branch), ('f', _Branch)]


_CONVS = ('l_conv1', 'l_conv2', 'l_conv3', 'l_conv4', 'r_conv1', 'r_conv2', 'r_conv3', 'r_conv4')


def _f32(t, what):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise L.GeobiError('%s: expected a float32 tensor on the device' % what)
    return t if t.is_contiguous() else t.contiguous()


def _pack_params(net):
    """(struct, tensors kept alive).  72 pointers; rebuilt per call (~10 us) so .to() / load_state_dict stay safe."""
    p, keep = _Params(), []

    def ptr(t, what):
        t = _f32(t.detach(), what)
        keep.append(t)
        return t.data_ptr()
    for gname in ('gnn_v', 'gnn_f'):
        g, mod = getattr(p, gname), getattr(net, gname)
        for i, cname in enumerate(_CONVS):
            c = getattr(mod, cname)
            g.conv[i].lin_w, g.conv[i].u_w = ptr(c.lin.weight, cname), ptr(c.u.weight, cname)
            g.conv[i].c, g.conv[i].bias = ptr(c.c, cname), ptr(c.bias, cname)
    for fc in ('fc_v1', 'fc_v2', 'fc_f1', 'fc_f2'):
        m = getattr(net, fc)
        setattr(p, fc + '_w', ptr(m.weight, fc))
        setattr(p, fc + '_b', ptr(m.bias, fc))
    p.force_depth = 1 if net.force_depth else 0
    p.pool_mean = 1 if net.gnn_v.pooling1.pool_type == 'mean' else 0
    return p, keep


def _param_order(net):
    """The parameters in the field order of geobi_net_params_t."""
    out = []
    for gname in ('gnn_v', 'gnn_f'):
        mod = getattr(net, gname)
        for cname in _CONVS:
            c = getattr(mod, cname)
            out += [c.lin.weight, c.u.weight, c.c, c.bias]
    for fc in ('fc_v1', 'fc_v2', 'fc_f1', 'fc_f2'):
        m = getattr(net, fc)
        out += [m.weight, m.bias]
    return out


def _pack_pointers(tensors, net):
    """A geobi_net_params_t over `tensors` (in _param_order)."""
    p = _Params()
    it = iter(tensors)
    for gname in ('gnn_v', 'gnn_f'):
        g = getattr(p, gname)
        for i in range(8):
            g.conv[i].lin_w, g.conv[i].u_w = next(it).data_ptr(), next(it).data_ptr()
            g.conv[i].c, g.conv[i].bias = next(it).data_ptr(), next(it).data_ptr()
    for fc in ('fc_v1', 'fc_v2', 'fc_f1', 'fc_f2'):
        setattr(p, fc + '_w', next(it).data_ptr())
        setattr(p, fc + '_b', next(it).data_ptr())
    p.force_depth = 1 if net.force_depth else 0
    p.pool_mean = 1 if net.gnn_v.pooling1.pool_type == 'mean' else 0
    return p


def supported(net):
    """The fast path covers the network as the reference builds it (edge_weight_type 10, two matching steps, no
    injected cluster vectors); everything else runs module by module."""
    for g in (net.gnn_v, net.gnn_f):
        for pl in (g.pooling1, g.pooling2):
            if pl.edge_weight_type != 10 or pl.pool_step != 2 or pl.graclus_fn is not None:
                return False
    return net.gnn_v.pooling1.pool_type == net.gnn_f.pooling1.pool_type


def _check_inputs(net, data_v, data_f, fv32, dd, tensors):
    """The library reads V / F rows through raw pointers: a malformed bag must raise here, not fault on the device
    (the module path reports the same mismatches through FeastConvFn / FaceGeomFn)."""
    V, F = data_v.x.shape[0], data_f.x.shape[0]
    dev = data_v.x.device
    if tuple(fv32.shape) != (F, 3):
        raise L.GeobiError('fv_indices has shape %s; the facet graph has %d nodes: expected (%d, 3)'
                           % (tuple(fv32.shape), F, F))
    if data_f.x.device != dev or fv32.device != dev:
        raise L.GeobiError('data_v and data_f live on different devices (%s, %s)' % (dev, data_f.x.device))
    if net.force_depth:
        if dd is None or dd.dim() != 2 or dd.shape[0] != V or dd.shape[1] != 3 or dd.device != dev:
            raise L.GeobiError('force_depth: depth_direction must be a [%d, 3] tensor on %s' % (V, dev))
    for t in tensors:
        if t.device != dev:
            raise L.GeobiError('parameters live on %s, the data on %s' % (t.device, dev))


def _level0(data, keep):
    g = data.graph(data.x.shape[0])
    if not g.symmetric or g.E == 0:
        return None
    w = getattr(data, 'edge_weight', None)
    if w is None:
        return None
    w = _f32(g.weights_sorted(w), 'edge_weight')
    row = g.ensure_rows()
    keep += [g, w, row]
    lv = _Level0()
    lv.N, lv.E = g.N, g.E
    lv.rowptr, lv.col, lv.row, lv.weight = g.rowptr_out.data_ptr(), g.col_out.data_ptr(), row.data_ptr(), w.data_ptr()
    return lv


# One grow-only arena per device, reused by every call: a fresh ~0.5 GB block per pass costs a device allocation
# whenever the batch shape changes (patch unions of different sizes).  Results are therefore COPIED out (two small
# tensors); what the pooling modules expose afterwards (unpool index, cluster vectors) stays in the arena and, like
# any module state of "the last forward", is valid until the next one.
_ARENAS = {}


def _arena_key(dev):
    """Arena and pass generation are per (device, stream): passes on one stream reuse the block in stream order; a pass
    on ANOTHER stream (a second host thread running its own inference pipeline) gets its own block."""
    return (dev.type, dev.index, L.stream())


def _arena(nbytes, dev):
    key = _arena_key(dev)
    a = _ARENAS.get(key)
    if a is None or a.numel() < nbytes:
        _ARENAS[key] = a = None                      # release before growing
        _ARENAS[key] = a = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=dev)
    return a


def _views(arena, off, n, dtype):
    nbytes = n * 4
    return arena[off:off + nbytes].view(dtype)


def _results(arena, out, V, F):
    """The two outputs copied out of the arena (which is released / overwritten by the next pass): the library places
    normals right behind verts, so one copy serves both."""
    lo, hi = int(out.verts_off), int(out.normals_off) + F * 12
    if 0 <= int(out.normals_off) - (lo + V * 12) <= 4096:
        both = arena[lo:hi].clone()
        verts = both[:V * 12].view(torch.float32).view(V, 3)
        normals = both[int(out.normals_off) - lo:].view(torch.float32).view(F, 3)
        return verts, normals
    return (_views(arena, out.verts_off, V * 3, torch.float32).view(V, 3).clone(),
            _views(arena, out.normals_off, F * 3, torch.float32).view(F, 3).clone())


def forward(net, data_v, data_f):
    """-> (verts [V,3], normals [F,3]) or None (not covered: the caller runs the module-by-module path).  Sets the
    module-surface side effects of a forward that callers read: PoolingLayer.unpooling_indices / last_clusters."""
    from .network import _fv_index
    if not (ENABLED and supported(net)):
        return None
    keep = []
    lv_v, lv_f = _level0(data_v, keep), _level0(data_f, keep)
    if lv_v is None or lv_f is None or data_v.x.shape[1] != 6 or data_f.x.shape[1] != 6:
        return None
    dev = data_v.x.device
    x_v, x_f = _f32(data_v.x, 'data_v.x'), _f32(data_f.x, 'data_f.x')
    fv32, _ = _fv_index(data_f, x_v.shape[0])
    dd = None
    if net.force_depth:
        if getattr(data_v, 'depth_direction', None) is None:
            raise L.GeobiError('force_depth: data_v.depth_direction is missing')
        dd = _f32(data_v.depth_direction, 'depth_direction')
    prm, pkeep = _pack_params(net)
    _check_inputs(net, data_v, data_f, fv32, dd, pkeep)
    lib = L.lib()
    nbytes = L.size_query('geobi_net_forward_arena_bytes', lv_v.N, lv_v.E, lv_f.N, lv_f.E)
    out = _Out()
    for _ in range(2):
        arena = _arena(nbytes, dev)
        rc = lib.geobi_net_forward(ctypes.byref(prm), ctypes.byref(lv_v), ctypes.byref(lv_f), x_v.data_ptr(),
                                   x_f.data_ptr(), fv32.data_ptr(), None if dd is None else dd.data_ptr(),
                                   arena.data_ptr(), arena.numel(), ctypes.byref(out), L.stream())
        if rc != 3:                                   # GEOBI_NET_ARENA: retry once with twice what was needed
            break
        nbytes = max(2 * int(out.used_bytes), 2 * nbytes)
        STATS['arena_retry'] += 1
    STATS['calls'] += 1
    if rc == 2:                                       # GEOBI_NET_FALLBACK
        STATS['fallback'] += 1
        return None
    if rc != 0:
        L.check(rc, 'geobi_net_forward')
    V, F = lv_v.N, lv_f.N
    verts, normals = _results(arena, out, V, F)
    # NOT reproduced: the reference's forward rewrites its input bags (network.py:271,298,337 leave data_v.x /
    # data_f.x as intermediate activations; the module path leaves the l_conv1 outputs there).  No caller of the
    # reference reads them back (train_dual.py:203-208, test_dual.py:21), and copying two [N, 32] activations out of
    # the arena per pass would cost more than the heads.  The executor leaves data_v.x / data_f.x as given; the
    # coupled facet features [F, 12] of network.py:337 are at `xf_off` of the arena for a caller that wants them.
    key = _arena_key(dev)
    _GENERATION[key] = _GENERATION.get(key, 0) + 1
    _set_module_state(net, arena, out, (key, _GENERATION[key]))
    return verts, normals


# Inference passes share one arena per device, so what the pooling modules expose after a forward (views into it) is
# overwritten by the NEXT pass of any net on that device.  Copying a dozen small vectors out per pass would cost ~40 us
# of launches on a 1 ms pass; instead every pass bumps the device's generation and the modules refuse a stale read
# (PoolingLayer._fresh) -- read them before the next forward, or run with GEOBI_NET_EXECUTOR=0.
_GENERATION = {}


def is_current(guard):
    return guard is None or _GENERATION.get(guard[0]) == guard[1]


def _clear_module_state(net):
    """Drop the previous pass' views BEFORE the next arena is allocated (they would pin the old block)."""
    for mod in (net.gnn_v, net.gnn_f):
        for pl in (mod.pooling1, mod.pooling2):
            pl._unpool32 = pl._unpool64 = pl._unpool_index = pl._last_clusters32 = None
            pl._arena_guard = None


def _set_module_state(net, arena, out, guard=None):
    """What the pooling modules expose after a forward (net_util.py:156): composed unpool index, raw cluster vectors.
    guard: (device key, generation) for views into the shared inference arena, None for a training pass' own arena."""
    for gname, bo in (('gnn_v', out.v), ('gnn_f', out.f)):
        mod = getattr(net, gname)
        for l, pl in enumerate((mod.pooling1, mod.pooling2)):
            pl._arena_guard = guard
            pl._unpool32 = _views(arena, bo.unpool_off[l], bo.nodes[l], torch.int32)
            pl._unpool64 = None
            pl._unpool_index = None                  # the executor keeps its own inverse lists
            pl._last_clusters32 = [_views(arena, bo.cluster_off[l][t], bo.cluster_len[l][t], torch.int32) for t in range(2)]


# ----------------------------------------------------------------------------------- training
_LEARNED = {}       # (V, Ev, F, Ef) -> arena bytes that were enough for forward AND backward of that shape


def _learned_size(table, key, used_bytes):
    """The arena size later passes on the same level-0 shape take: 5 % over the exact need, rounded up to 64 MiB, and never
    smaller than what was learned before.  Different meshes of one shape differ by a fraction of a percent in their coarse
    sizes; a size that followed every pass up and down made the block allocator hand out a NEW multi-GB block whenever the
    request grew past the cached one (bench extra.fresh_batch: 4.2 ms per step became 7.8-10.3 until the sizes settled)."""
    q = 64 << 20
    want = (int(1.05 * int(used_bytes)) + (16 << 20) + q - 1) // q * q
    table[key] = max(table.get(key, 0), want)
    return table[key]

class Recorded(object):
    """A training forward recorded by the library: the arena with every saved buffer, the host-side record (handle)
    and what the backward needs from the Python side.  The record is released with the object."""

    def __init__(self, handle, arena, keep, net, corner):
        self.handle, self.arena, self.keep, self.net, self.corner = handle, arena, keep, net, corner

    def release(self):
        if self.handle:
            L.lib().geobi_net_release(ctypes.c_int64(self.handle))
            self.handle = 0

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def forward_train(net, data_v, data_f):
    """-> (verts, normals, Recorded) or None (outside the fast path: the caller records the op tape instead)."""
    from .network import _fv_index
    if not (ENABLED and supported(net)):
        return None
    params = _param_order(net)
    if not all(p.requires_grad for p in params) or data_v.x.requires_grad or data_f.x.requires_grad:
        return None
    keep = []
    lv_v, lv_f = _level0(data_v, keep), _level0(data_f, keep)
    if lv_v is None or lv_f is None or data_v.x.shape[1] != 6 or data_f.x.shape[1] != 6:
        return None
    gv, gf = data_v.graph().ensure_in(), data_f.graph().ensure_in()
    dev = data_v.x.device
    x_v, x_f = _f32(data_v.x, 'data_v.x'), _f32(data_f.x, 'data_f.x')
    fv32, corner = _fv_index(data_f, x_v.shape[0])
    dd = None
    if net.force_depth:
        if getattr(data_v, 'depth_direction', None) is None:
            raise L.GeobiError('force_depth: data_v.depth_direction is missing')
        dd = _f32(data_v.depth_direction, 'depth_direction')
    tensors = [_f32(p.detach(), 'parameter') for p in params]
    _check_inputs(net, data_v, data_f, fv32, dd, tensors)
    prm = _pack_pointers(tensors, net)
    keep += [tensors, x_v, x_f, fv32, dd, gv, gf]
    _clear_module_state(net)
    lib = L.lib()
    shape_key = (lv_v.N, lv_v.E, lv_f.N, lv_f.E)
    nbytes = _LEARNED.get(shape_key) or L.size_query('geobi_net_train_arena_bytes', *shape_key)
    out, handle = _Out(), ctypes.c_int64(0)
    for _ in range(4):
        arena = None                                   # release a too-small block before taking a larger one
        arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = lib.geobi_net_forward_train(ctypes.byref(prm), ctypes.byref(lv_v), ctypes.byref(lv_f), gv.pos_in.data_ptr(),
                                         gf.pos_in.data_ptr(), x_v.data_ptr(), x_f.data_ptr(), fv32.data_ptr(),
                                         None if dd is None else dd.data_ptr(), arena.data_ptr(), nbytes,
                                         ctypes.byref(out), ctypes.byref(handle), L.stream())
        if rc != 3:
            break
        used = int(out.used_bytes)       # mid-forward overflow: bytes so far; end-of-forward check: the exact total
        nbytes = int(1.25 * used) + (64 << 20) if used > nbytes else 2 * nbytes
        STATS['arena_retry'] += 1
    STATS['calls'] += 1
    if rc == 2:
        STATS['fallback'] += 1
        return None
    if rc != 0:
        L.check(rc, 'geobi_net_forward_train')
    # the library reports the exact need (forward + backward) of this mesh: later steps on the same shape take that
    _learned_size(_LEARNED, shape_key, out.used_bytes)
    STATS['train_arena_bytes'], STATS['train_need_bytes'] = nbytes, int(out.used_bytes)
    V, F = lv_v.N, lv_f.N
    # results are COPIED out (two small tensors): a view would pin the whole training arena (GBs) for as long as the
    # caller keeps the prediction, and the next step's arena could not reuse the block
    verts, normals = _results(arena, out, V, F)
    # the pooling modules' state stays a set of views into this pass' own arena (valid, but the block is held until
    # the next training forward drops them -- before it allocates, see _clear_module_state -- so never two arenas)
    _set_module_state(net, arena, out)
    return verts, normals, Recorded(handle.value, arena, keep, net, corner)


# Data-parallel overlap (parallel.GradBucket.all_reduce_mean_split): two torch.cuda.Event objects that the NEXT backward
# records where the facet half of the gradients is final (geobi_net_backward_facet_events).  Set here, not through the
# library from the caller's thread: autograd runs this backward on its own device thread, and the hook is per host thread.
FACET_EVENTS = None


def backward(rec, g_verts, g_normals, params):
    """Parameter gradients of a recorded forward, aligned with `params` (None where the kernels added straight into a
    direct-gradient bucket view, see ops._direct_grad)."""
    from .ops import _direct_grad
    net = rec.net
    ordered = _param_order(net)
    direct = [_direct_grad(p) for p in ordered]
    use_direct = all(d is not None for d in direct)
    grads = direct if use_direct else [torch.empty_like(p, memory_format=torch.contiguous_format) for p in ordered]
    gp = _pack_pointers(grads, net)
    cidx = rec.corner.get()
    gv = None if g_verts is None else _f32(g_verts, 'grad of verts')
    gn = None if g_normals is None else _f32(g_normals, 'grad of normals')
    if FACET_EVENTS is not None:
        L.lib().geobi_net_backward_facet_events(ctypes.c_void_p(FACET_EVENTS[0].cuda_event),
                                                ctypes.c_void_p(FACET_EVENTS[1].cuda_event))
    rc = L.lib().geobi_net_backward(ctypes.c_int64(rec.handle), None if gv is None else gv.data_ptr(),
                                    None if gn is None else gn.data_ptr(), ctypes.byref(gp), 1 if use_direct else 0,
                                    cidx.segptr.data_ptr(), cidx.members.data_ptr(), L.stream())
    if rc != 0:
        L.check(rc, 'geobi_net_backward')
    rec.keep.append((gv, gn, grads))        # alive until the record goes (stream-ordered reuse is safe after that)
    if use_direct:
        return [None] * len(params)
    by_id = {id(p): g for p, g in zip(ordered, grads)}
    return [by_id.get(id(p)) for p in params]


# ----------------------------------------------------------------------------------- mesh groups in flight together
class _Group(ctypes.Structure):
    """geobi_train_group_t (include/geobi_hip.h)."""
    _fields_ = [('gv', _F), ('gf', _F), ('pos_rev_v', _F), ('pos_rev_f', _F), ('x_v', _F), ('x_f', _F), ('fv', _F),
                ('depth_direction', _F), ('y_v', _F), ('y_f', _F), ('w_v', _F), ('w_f', _F),
                ('scale_v', ctypes.c_float), ('scale_n', ctypes.c_float), ('corner_segptr', _F), ('corner_members', _F),
                ('arena', _F), ('arena_bytes', ctypes.c_size_t), ('grads', _Params), ('grad_flat', _F),
                ('grad_count', ctypes.c_int64), ('losses', _F), ('stream', _F), ('out', _Out), ('rc', ctypes.c_int32),
                ('error', ctypes.c_char * 252)]


_KINDS = {'L1': 0, 'L2': 1}


class TrainGroups(object):
    """The meshes of one optimiser step as several GROUPS in flight together (geobi_net_train_groups).

    The reference walks the meshes of a batch one after another -- forward, ``loss / batch_size``, backward, optimiser
    step every ``batch_size`` meshes (/root/reference/code/train_dual.py:199-218).  Meshes are independent, so the
    iterations of that loop can overlap: every group (a disjoint-union pair ``(data_v, data_f)`` of one or more meshes)
    runs forward -> loss -> backward on its own stream, driven by its own host thread inside the library, with its own
    arena and gradient bucket; the buckets are added up in group order into ``bucket.flat`` (bit-reproducible).  One
    group's pooling chains and size reads then run under the other groups' FeaSt kernels.

    ``step()`` returns the per-group losses ``[G, 2]`` (each group's SHARE of loss_v / loss_n: they add up to the step's
    losses as ``parallel.batched_losses`` over the union of all groups gives them) and leaves the summed gradient in
    ``bucket.flat``; the caller all-reduces and steps the optimiser as after ``loss.backward()``.  Cases outside the
    executor's fast path run the same step group by group through the module path."""

    def __init__(self, net, bucket, loss_v='L1', loss_n='L1', v_scale=1.0, n_scale=1.0):
        if loss_v not in _KINDS or loss_n not in _KINDS:
            raise NotImplementedError('TrainGroups: L1 / L2 losses only (network.loss_v / loss_n)')
        self.net, self.bucket = net, bucket
        self.kinds = (loss_v, loss_n)
        self.scales = (float(v_scale), float(n_scale))
        self.groups, self._prep, self.losses = [], [], None
        self.sequential_steps = 0
        order = _param_order(net)
        if not all(p.requires_grad for p in order) or len(bucket.params) != len(order):
            raise L.GeobiError('TrainGroups: every parameter of the network must require a gradient and live in the bucket')

    # ---- per-group resources
    def _group_bucket(self, dev):
        """A flat gradient buffer laid out like bucket.flat, and the struct of pointers into it (struct order)."""
        flat = torch.zeros_like(self.bucket.flat)
        views, off = {}, 0
        for p in self.bucket.params:
            n = p.numel()
            views[id(p)] = flat[off:off + n]
            off += n
        return flat, _pack_pointers([views[id(p)] for p in _param_order(self.net)], self.net)

    def set_groups(self, groups):
        """groups: list of (data_v, data_f) pairs on the device (each a single mesh or a union batch with mesh_ptr)."""
        from .network import _fv_index
        from .parallel import _mesh_weights
        if not 1 <= len(groups) <= 8:
            raise ValueError('TrainGroups: 1..8 groups')
        net = self.net
        n_mesh = []
        for dv, df in groups:
            ptr = getattr(dv, 'mesh_ptr', None)
            n_mesh.append(1 if ptr is None or ptr.numel() <= 2 else ptr.numel() - 1)
        total = float(sum(n_mesh))
        old = self._prep
        self.groups, self._prep, self._call = list(groups), [], None
        for k, (dv, df) in enumerate(groups):
            keep = []
            lv_v, lv_f = _level0(dv, keep), _level0(df, keep)
            if lv_v is None or lv_f is None or dv.x.shape[1] != 6 or df.x.shape[1] != 6 or dv.y is None or df.y is None:
                raise L.GeobiError('TrainGroups: group %d is not a symmetric weighted mesh pair with targets' % k)
            gv, gf = dv.graph().ensure_in(), df.graph().ensure_in()
            dev = dv.x.device
            x_v, x_f = _f32(dv.x, 'data_v.x'), _f32(df.x, 'data_f.x')
            y_v, y_f = _f32(dv.y, 'data_v.y'), _f32(df.y, 'data_f.y')
            fv32, corner = _fv_index(df, x_v.shape[0])
            cidx = corner.get()
            dd = None
            if net.force_depth:
                if getattr(dv, 'depth_direction', None) is None:
                    raise L.GeobiError('force_depth: data_v.depth_direction is missing')
                dd = _f32(dv.depth_direction, 'depth_direction')
            if tuple(y_v.shape) != (x_v.shape[0], 3) or tuple(y_f.shape) != (x_f.shape[0], 3):
                raise L.GeobiError('TrainGroups: targets must be [V, 3] and [F, 3]')
            _check_inputs(net, dv, df, fv32, dd, [p.detach() for p in _param_order(net)])
            w_v, w_f = _mesh_weights(dv), _mesh_weights(df)
            g = _Group()
            g.gv, g.gf = ctypes.addressof(lv_v), ctypes.addressof(lv_f)
            g.pos_rev_v, g.pos_rev_f = gv.pos_in.data_ptr(), gf.pos_in.data_ptr()
            g.x_v, g.x_f, g.fv = x_v.data_ptr(), x_f.data_ptr(), fv32.data_ptr()
            g.depth_direction = None if dd is None else dd.data_ptr()
            g.y_v, g.y_f = y_v.data_ptr(), y_f.data_ptr()
            g.w_v = None if w_v is None else w_v.data_ptr()
            g.w_f = None if w_f is None else w_f.data_ptr()
            share = n_mesh[k] / total
            g.scale_v, g.scale_n = self.scales[0] * share, self.scales[1] * share
            g.corner_segptr, g.corner_members = cidx.segptr.data_ptr(), cidx.members.data_ptr()
            # streams, buckets and arenas are per SLOT and survive a change of meshes
            res = old[k]['res'] if k < len(old) else None
            if res is None:
                flat, gp = self._group_bucket(dev)
                res = {'stream': torch.cuda.Stream(device=dev), 'flat': flat, 'gp': gp, 'arena': None}
            shape_key = (lv_v.N, lv_v.E, lv_f.N, lv_f.E)
            self._prep.append({'g': g, 'keep': keep + [lv_v, lv_f, gv, gf, x_v, x_f, y_v, y_f, fv32, cidx, dd, w_v, w_f],
                               'res': res, 'shape': shape_key, 'dev': dev})
        dev = self._prep[0]['dev']
        if self.losses is None or self.losses.shape[0] != len(groups):
            self.losses = torch.zeros(len(groups), 2, dtype=torch.float32, device=dev)
        return self

    def _arena(self, pr, grow=None):
        res = pr['res']
        need = grow or _LEARNED_GROUP.get(pr['shape']) or L.size_query('geobi_net_train_arena_bytes', *pr['shape'])
        if res['arena'] is None or res['arena'].numel() < need:
            res['arena'] = None
            res['arena'] = torch.empty(int(need), dtype=torch.uint8, device=pr['dev'])
        return res['arena']

    def step(self):
        if not self._prep:
            raise L.GeobiError('TrainGroups.step: set_groups first')
        net, n = self.net, len(self._prep)
        if not (ENABLED and supported(net)):
            return self._sequential()
        # the parameter struct and the group array are rebuilt only when a pointer they hold has moved (the flat parameter
        # of FlatParameters never does): ~100 us of interpreter work per step otherwise, with every group waiting for it
        order = _param_order(net)
        sig = (order[0].data_ptr(), order[-1].data_ptr(), self.bucket.flat.data_ptr(),
               tuple(0 if pr['res']['arena'] is None else pr['res']['arena'].data_ptr() for pr in self._prep))
        cached = getattr(self, '_call', None)
        lib = L.lib()
        for attempt in range(4):
            if cached is None or cached[0] != sig or attempt > 0:
                tensors = [_f32(p.detach(), 'parameter') for p in order]
                prm = _pack_pointers(tensors, net)
                arr = (_Group * n)()
                for k, pr in enumerate(self._prep):
                    g, res = pr['g'], pr['res']
                    arena = self._arena(pr)
                    g.arena, g.arena_bytes = arena.data_ptr(), arena.numel()
                    g.grads, g.grad_flat, g.grad_count = res['gp'], res['flat'].data_ptr(), res['flat'].numel()
                    g.losses = self.losses.data_ptr() + 8 * k
                    g.stream = res['stream'].cuda_stream
                    arr[k] = g
                sig = sig[:3] + (tuple(pr['res']['arena'].data_ptr() for pr in self._prep),)
                cached = self._call = (sig, tensors, prm, arr)
            _, tensors, prm, arr = cached
            rc = lib.geobi_net_train_groups(ctypes.byref(prm), ctypes.cast(arr, ctypes.c_void_p), n,
                                            _KINDS[self.kinds[0]], _KINDS[self.kinds[1]], self.bucket.flat.data_ptr(),
                                            self.bucket.flat.numel(), L.stream())
            STATS['calls'] += 1
            if rc != 3:
                break
            STATS['arena_retry'] += 1
            for k, pr in enumerate(self._prep):            # every group reports what it needs
                if arr[k].rc == 3:
                    used = int(arr[k].out.used_bytes)
                    have = pr['res']['arena'].numel()
                    self._arena(pr, grow=int(1.25 * used) + (64 << 20) if used > have else 2 * have)
        if rc == 2:
            STATS['fallback'] += 1
            return self._sequential()
        if rc != 0:
            L.check(rc, 'geobi_net_train_groups')
        for k, pr in enumerate(self._prep):
            _learned_size(_LEARNED_GROUP, pr['shape'], arr[k].out.used_bytes)
            pr['out'] = arr[k].out
        return self.losses

    def prediction(self, k):
        """(verts, normals) of group k from the last step: VIEWS into the group's arena, valid until the next step."""
        pr = self._prep[k]
        out, arena = pr['out'], pr['res']['arena']
        V, F = pr['shape'][0], pr['shape'][2]
        return (_views(arena, out.verts_off, V * 3, torch.float32).view(V, 3),
                _views(arena, out.normals_off, F * 3, torch.float32).view(F, 3))

    def _sequential(self):
        """The same step group by group through the network's autograd node (module path or single-stream executor)."""
        from .parallel import batched_losses
        self.sequential_steps += 1
        self.bucket.zero()
        rows = []
        for pr, (dv, df) in zip(self._prep, self.groups):
            vp, npred, _ = self.net((dv.shallow_copy(), df.shallow_copy()))
            lv, ln = batched_losses(vp, npred, dv, df, self.kinds[0], self.kinds[1])
            lv, ln = lv * float(pr['g'].scale_v), ln * float(pr['g'].scale_n)
            (lv + ln).backward()
            rows.append(torch.stack([lv.detach(), ln.detach()]))
        self.losses = torch.stack(rows)
        return self.losses


_LEARNED_GROUP = {}
