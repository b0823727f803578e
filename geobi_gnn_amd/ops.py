"""torch.autograd.Function wrappers over the C ABI (include/geobi_hip.h).

Each Function replaces one third-party op of the reference's hot path; the replaced call
site is cited in the docstring.  Tensors are allocated by PyTorch's caching allocator and
handed to the library as borrowed device pointers on torch's current HIP stream.
"""
import os

import torch
from torch.autograd import Function

from . import _lib as L

LEAK = 0.2
HP = 12   # GEOBI_HEAD_STRIDE
# GEOBI_FUSED=0: separate aggregation + GEMM kernels with the aggregated rows z [N, 9 Cin] written to HBM
# (the round-1 path, kept for A/B timing and as the second implementation the fused kernels are tested against)
FUSED = os.environ.get('GEOBI_FUSED', '1') == '1'


# ----------------------------------------------------------------------------- op tape
# DualGNN.forward issues ~70 differentiable ops per step.  Going through torch.autograd for each costs
# ~15 us of host time per node (Function.apply, graph bookkeeping, engine thread hops) -- more than
# the kernels of the coarse levels take.  Inside DualGNN the same Function classes are therefore
# driven through a minimal reverse-mode tape: the whole network is ONE autograd node
# (network.DualGNNFn) and the tape replays the per-op `backward`s in reverse order.  Outside of it
# (FeaStConv / PoolingLayer used on their own) the ops are ordinary autograd Functions.
class _TapeCtx(object):
    __slots__ = ('needs_input_grad', 'saved_tensors', '__dict__')

    def __init__(self, needs):
        self.needs_input_grad = needs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class Tape(object):
    """Reverse-mode tape over the Function classes of this module (forward called in no-grad mode)."""

    def __init__(self, record=True):
        self.record = record
        self.nodes = []
        self.live = set()              # ids of intermediate tensors that carry a gradient

    def needs(self, a):
        return torch.is_tensor(a) and a.is_floating_point() and (a.requires_grad or id(a) in self.live)

    def apply(self, fn, *args):
        needs = tuple(self.needs(a) for a in args)
        ctx = _TapeCtx(needs)
        out = fn.forward(ctx, *args)
        if self.record and any(needs):
            self.nodes.append((fn, ctx, args, out))
            self.live.add(id(out))
        return out

    def backward(self, seeds, clear=True):
        """seeds: {id(tensor): grad}.  Returns {id(leaf tensor): grad} for leaves with requires_grad.
        clear=False keeps the recorded nodes (and the tensors they saved) alive for the caller to drop."""
        grads = dict(seeds)
        leaf = {}
        for fn, ctx, args, out in reversed(self.nodes):
            g = grads.pop(id(out), None)
            if g is None:
                continue
            gins = fn.backward(ctx, g)
            for a, gi in zip(args, gins):
                if gi is None or not torch.is_tensor(a):
                    continue
                k = id(a)
                if k in self.live:
                    grads[k] = gi if k not in grads else grads[k] + gi
                elif a.requires_grad:
                    leaf[k] = gi if k not in leaf else leaf[k] + gi
        if clear:
            self.nodes = []
        return leaf


_ACTIVE_TAPE = [None]

# Deferred side-stream join (geobi_side_defer): while a list is installed here, backward ops park every
# buffer the side stream may still be reading (workspaces, incoming gradients) in it; whoever installed
# it joins the side stream and only then drops the list.
_SIDE_KEEP = [None]


class deferred_side_join(object):
    """with deferred_side_join(): ... -- weight-gradient GEMMs of all backward calls inside queue on the
    library's side stream without per-call joins; one join on exit."""

    def __init__(self, enable=True):
        self.enable = enable

    def __enter__(self):
        self.prev = _SIDE_KEEP[0]
        if self.enable and self.prev is None:
            _SIDE_KEEP[0] = []
            L.call('geobi_side_defer', 1)
        return self

    def __exit__(self, *exc):
        if self.enable and self.prev is None:
            try:
                L.call('geobi_side_defer', 0)
                L.call('geobi_side_join', L.stream())
            finally:
                _SIDE_KEEP[0] = None
        return False


def apply_op(fn, *args):
    """Run a Function through the active tape (inside DualGNN) or through torch.autograd."""
    tape = _ACTIVE_TAPE[0]
    return tape.apply(fn, *args) if tape is not None else fn.apply(*args)


class use_tape(object):
    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        self.prev = _ACTIVE_TAPE[0]
        _ACTIVE_TAPE[0] = self.tape
        return self.tape

    def __exit__(self, *exc):
        _ACTIVE_TAPE[0] = self.prev
        return False


def _direct_grad(p):
    """Gradient buffer the kernels may write into directly.

    parallel.GradBucket(direct=True) marks parameters whose ``.grad`` is a persistent view of the
    flat bucket.  The backward kernels then ADD the gradient straight into that view (`accumulate`
    argument of geobi_feast_bwd / geobi_head_bwd) and autograd receives ``None`` -- this removes one
    accumulate kernel per parameter (76 per step) and keeps autograd's semantics: several backward
    passes between two ``bucket.zero()`` calls sum up (the reference's gradient accumulation,
    train_dual.py:211-218), and so do several uses of one parameter in a forward."""
    if not getattr(p, '_geobi_direct_grad', False):
        return None
    if p.grad is None or not p.grad.is_contiguous() or getattr(p, '_geobi_grad_ptr', None) != p.grad.data_ptr():
        raise L.GeobiError('a direct-gradient parameter lost its bucket view (optimizer.zero_grad(set_to_none=True) or '
                           'an assignment to .grad): zero the gradients with GradBucket.zero() instead')
    bucket = getattr(p, '_geobi_bucket', None)
    owner = None if bucket is None else bucket.owner
    if owner is not None and (owner.grad is None or owner.grad.data_ptr() != bucket.flat.data_ptr()):
        # FlatParameters: the optimizer steps ONE flat parameter.  zero_grad(set_to_none=True) on it leaves the
        # per-parameter views intact, the kernels would keep adding into the bucket and optimizer.step() would skip
        # the parameter without a word
        raise L.GeobiError('the flat parameter lost its gradient bucket (optimizer.zero_grad(set_to_none=True) or an '
                           'assignment to .grad): zero the gradients with GradBucket.zero() instead')
    return p.grad


def _f32c(t):
    if t.dtype != torch.float32:
        raise L.GeobiError('expected float32, got %s' % t.dtype)
    return t.contiguous()


# --------------------------------------------------------------------------- FeaSt conv
class FeastConvFn(Function):
    """torch_geometric.nn.FeaStConv forward/backward (network.py:271-299), optionally fused with
    the following leaky_relu (``slope``) and the skip concatenation (input given as xa | xb)."""

    @staticmethod
    def forward(ctx, xa, xb, lin_w, u_w, c, bias, graph, slope):
        L.require_device(xa, 'x')
        g = graph.ensure_in()
        ctx.param_refs = (lin_w, u_w, c, bias)
        xa = _f32c(xa)
        xb = None if xb is None else _f32c(xb)
        lin_w, u_w, c, bias = _f32c(lin_w), _f32c(u_w), _f32c(c), _f32c(bias)
        N, Ca = xa.shape
        Cb = 0 if xb is None else xb.shape[1]
        Cin, Cout = Ca + Cb, bias.shape[0]
        if N != g.N:
            raise L.GeobiError('FeaStConv: x has %d rows but the graph has %d nodes' % (N, g.N))
        if lin_w.shape != (9 * Cout, Cin) or u_w.shape != (9, Cin) or c.shape != (9,):
            raise L.GeobiError('FeaStConv: parameter shapes do not match heads=9, in=%d, out=%d' % (Cin, Cout))
        dev = xa.device
        lib = L.lib()
        ldz = L.size_query('geobi_feast_ldz', Cin)
        out = torch.empty((N, Cout), dtype=torch.float32, device=dev)
        p = torch.empty((N, HP), dtype=torch.float32, device=dev)
        z = None if FUSED else torch.empty((N, ldz), dtype=torch.float32, device=dev)
        # packed weights (Wf | W'), reused by the backward; not kept when nothing needs a gradient
        wf = None
        if any(ctx.needs_input_grad):
            wf = torch.empty(L.size_query('geobi_feast_wpack_floats', Cin, Cout), dtype=torch.float32, device=dev)
        ws = L.workspace(L.size_query('geobi_feast_fwd_ws_bytes', N, Cin, Cout), dev)
        L.call('geobi_feast_fwd', L.ptr(xa), L.ptr(xb), Ca, Cb, N, g.E, L.ptr(g.rowptr_in), L.ptr(g.col_in),
               L.ptr(lin_w), L.ptr(u_w), L.ptr(c), L.ptr(bias), Cout, float(slope), L.ptr(out), L.ptr(p), L.ptr(z),
               L.ptr(wf), L.ptr(ws), ws.numel(), L.stream())
        ctx.graph, ctx.slope, ctx.has_b, ctx.has_wf, ctx.has_z = g, float(slope), xb is not None, wf is not None, z is not None
        ctx.save_for_backward(xa, xb if xb is not None else xa, lin_w, u_w, c, out, p, z if z is not None else p,
                              wf if wf is not None else p)
        return out

    @staticmethod
    def backward(ctx, gout):
        xa, xb, lin_w, u_w, c, out, p, z, wf = ctx.saved_tensors
        if not ctx.has_wf:
            wf = None
        if not ctx.has_z:
            z = None
        g = ctx.graph
        if not ctx.has_b:
            xb = None
        N, Ca = xa.shape
        Cb = 0 if xb is None else xb.shape[1]
        Cin, Cout = Ca + Cb, out.shape[1]
        dev = xa.device
        gout = _f32c(gout)
        need_dx = ctx.needs_input_grad[0] or (xb is not None and ctx.needs_input_grad[1])
        dxa = torch.empty_like(xa) if need_dx else None
        dxb = torch.empty_like(xb) if (need_dx and xb is not None) else None
        direct = [_direct_grad(p_) for p_ in ctx.param_refs]
        if all(d is not None for d in direct):
            dlin, du, dc, dbias = direct
            ret = (None, None, None, None)
        else:
            dlin, du, dc = torch.empty_like(lin_w), torch.empty_like(u_w), torch.empty_like(c)
            dbias = torch.empty(Cout, dtype=torch.float32, device=dev)
            ret = (dlin, du, dc, dbias)
        ws = L.workspace(L.size_query('geobi_feast_bwd_ws_bytes', N, g.E, Cin, Cout), dev)
        L.call('geobi_feast_bwd', L.ptr(xa), L.ptr(xb), Ca, Cb, N, g.E, L.ptr(g.rowptr_in), L.ptr(g.col_in),
               L.ptr(g.rowptr_out), L.ptr(g.col_out), L.ptr(g.pos_in), L.ptr(lin_w), L.ptr(u_w), L.ptr(c), Cout,
               ctx.slope, L.ptr(out), L.ptr(gout), L.ptr(p), L.ptr(z), L.ptr(wf), L.ptr(dxa), L.ptr(dxb), L.ptr(dlin),
               L.ptr(du), L.ptr(dc), L.ptr(dbias), 1 if ret[0] is None else 0, L.ptr(ws), ws.numel(), L.stream())
        if _SIDE_KEEP[0] is not None:
            _SIDE_KEEP[0].append((ws, gout, dlin, du, dc, dbias))
        return (dxa, dxb) + ret + (None, None)


def feast_conv(x, graph, lin_w, u_w, c, bias, slope=1.0, x2=None):
    return apply_op(FeastConvFn, x, x2, lin_w, u_w, c, bias, graph, slope)


# ------------------------------------------------------------------ inverse lists / pooling
class SegmentIndex(object):
    """Inverse lists ``segment -> members`` of an int32 segment-id vector (sorted, deterministic)."""

    def __init__(self, seg32, nseg, build=True):
        L.require_device(seg32, 'segment ids')
        self.seg = seg32.contiguous()
        self.n = int(seg32.shape[0])
        self.nseg = int(nseg)
        dev = seg32.device
        self.segptr = torch.empty(self.nseg + 1, dtype=torch.int32, device=dev)
        self.members = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)[:self.n]
        if build:       # general path: radix sort of (segment, member) keys
            ws = L.workspace(L.size_query('geobi_segment_csr_ws_bytes', self.n), dev)
            L.call('geobi_segment_csr', L.ptr(self.seg), self.n, self.nseg, L.ptr(self.segptr),
                   L.ptr(self.members), L.ptr(ws), ws.numel(), L.stream())

    @staticmethod
    def view(seg32, nseg, segptr, members):
        """Lists that already exist (slices of arrays built with an upper bound on the sizes)."""
        self = SegmentIndex.__new__(SegmentIndex)
        self.seg, self.n, self.nseg = seg32, int(seg32.shape[0]), int(nseg)
        self.segptr, self.members = segptr, members
        return self

    @staticmethod
    def from_matching(cnew32, raw32, nseg):
        """Sort-free lists for a matching (clusters of <= 2 nodes, raw id = smaller member)."""
        self = SegmentIndex(cnew32, nseg, build=False)
        raw32 = raw32.contiguous()
        ws = L.workspace(L.size_query('geobi_segment_pairs_ws_bytes', self.nseg), cnew32.device)
        L.call('geobi_segment_csr_pairs', L.ptr(self.seg), L.ptr(raw32), self.n, self.nseg, L.ptr(self.segptr),
               L.ptr(self.members), L.ptr(ws), ws.numel(), L.stream())
        return self

    def narrow(self, nseg):
        """Lists built with an upper bound on the segment count: keep the first `nseg` segments."""
        self.nseg = int(nseg)
        self.segptr = self.segptr[:self.nseg + 1]
        return self

    @staticmethod
    def compose(first, second, composed_seg32):
        """Lists of fine -> coarse for `composed_seg32 = second.seg[first.seg]`."""
        self = SegmentIndex(composed_seg32, second.nseg, build=False)
        ws = L.workspace(L.size_query('geobi_segment_pairs_ws_bytes', self.nseg), composed_seg32.device)
        L.call('geobi_segment_csr_compose', L.ptr(first.segptr), L.ptr(first.members), L.ptr(second.segptr),
               L.ptr(second.members), self.nseg, self.n, L.ptr(self.segptr), L.ptr(self.members), L.ptr(ws),
               ws.numel(), L.stream())
        return self


class SegmentMaxFn(Function):
    """torch_scatter.scatter(x, cluster, dim=0, reduce='max') (net_util.py:134)."""

    @staticmethod
    def forward(ctx, x, sidx):
        x = _f32c(x)
        C = x.shape[1]
        out = torch.empty((sidx.nseg, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((sidx.nseg, C), dtype=torch.int32, device=x.device)
        L.call('geobi_segment_max_fwd', L.ptr(x), C, L.ptr(sidx.segptr), L.ptr(sidx.members), sidx.nseg, L.ptr(out),
               L.ptr(arg), L.stream())
        ctx.save_for_backward(arg)
        ctx.n_fine, ctx.sidx = x.shape[0], sidx
        return out

    @staticmethod
    def backward(ctx, gout):
        (arg,) = ctx.saved_tensors
        gout = _f32c(gout)
        nseg, C = gout.shape
        gx = torch.empty((ctx.n_fine, C), dtype=torch.float32, device=gout.device)
        L.call('geobi_segment_max_bwd', L.ptr(gout), L.ptr(arg), L.ptr(ctx.sidx.seg), C, nseg, ctx.n_fine, L.ptr(gx),
               L.stream())
        return gx, None


class SegmentMeanFn(Function):
    """torch_scatter.scatter(..., reduce='mean') (net_util.py:132; pool_pos :136)."""

    @staticmethod
    def forward(ctx, x, sidx):
        x = _f32c(x)
        C = x.shape[1]
        out = torch.empty((sidx.nseg, C), dtype=torch.float32, device=x.device)
        L.call('geobi_segment_sum', L.ptr(x), C, L.ptr(sidx.segptr), L.ptr(sidx.members), sidx.nseg, 1, L.ptr(out),
               L.stream())
        ctx.sidx = sidx
        return out

    @staticmethod
    def backward(ctx, gout):
        sidx = ctx.sidx
        gout = _f32c(gout)
        C = gout.shape[1]
        gx = torch.empty((sidx.n, C), dtype=torch.float32, device=gout.device)
        L.call('geobi_segment_mean_bwd', L.ptr(gout), L.ptr(sidx.seg), L.ptr(sidx.segptr), C, sidx.n, L.ptr(gx),
               L.stream())
        return gx, None


class UnpoolFn(Function):
    """PoolingLayer.unpooling: ``x[unpooling_indices]`` (net_util.py:242-245); the backward is a
    sorted-segment sum through the inverse lists instead of an atomic index_add."""

    @staticmethod
    def forward(ctx, x, sidx):
        x = _f32c(x)
        C = x.shape[1]
        out = torch.empty((sidx.n, C), dtype=torch.float32, device=x.device)
        L.call('geobi_gather_rows', L.ptr(x), L.ptr(sidx.seg), C, sidx.n, L.ptr(out), L.stream())
        ctx.sidx = sidx
        return out

    @staticmethod
    def backward(ctx, gout):
        sidx = ctx.sidx
        gout = _f32c(gout)
        C = gout.shape[1]
        gx = torch.empty((sidx.nseg, C), dtype=torch.float32, device=gout.device)
        L.call('geobi_segment_sum', L.ptr(gout), C, L.ptr(sidx.segptr), L.ptr(sidx.members), sidx.nseg, 0,
               L.ptr(gx), L.stream())
        return gx, None


# ----------------------------------------------------------------------- geometry / heads
class FaceGeomFn(Function):
    """network.py:335-337 + data_util.computer_face_normal: x_f = cat(x_f, centroid, unit normal)."""

    @staticmethod
    def forward(ctx, verts, xf, fv32, corner_index):
        verts, xf = _f32c(verts), _f32c(xf)
        F = fv32.shape[0]
        out = torch.empty((F, 12), dtype=torch.float32, device=verts.device)
        L.call('geobi_face_geom_fwd', L.ptr(verts), L.ptr(fv32), L.ptr(xf), xf.shape[1], F, L.ptr(out), L.stream())
        ctx.save_for_backward(verts, fv32)
        ctx.corner_index = corner_index
        return out

    @staticmethod
    def backward(ctx, gout):
        verts, fv32 = ctx.saved_tensors
        cidx = ctx.corner_index.get() if hasattr(ctx.corner_index, 'get') else ctx.corner_index
        gout = _f32c(gout)
        F = fv32.shape[0]
        cg = torch.empty((3 * F, 3), dtype=torch.float32, device=gout.device)
        L.call('geobi_face_geom_bwd', L.ptr(verts), L.ptr(fv32), L.ptr(gout), F, L.ptr(cg), L.stream())
        gv = torch.empty((cidx.nseg, 3), dtype=torch.float32, device=gout.device)
        L.call('geobi_segment_sum', L.ptr(cg), 3, L.ptr(cidx.segptr), L.ptr(cidx.members), cidx.nseg, 0, L.ptr(gv),
               L.stream())
        return gv, None, None, None


class HeadFn(Function):
    """fc2(leaky_relu(fc1 x)) + finish (network.py:324-332 vertex head, :340-343 face head).

    Fused kernels (Cin = 32, K = 1024): the [N, 1024] hidden activation lives in MFMA accumulators
    only -- forward and backward (recomputed) -- and is never written to HBM.  Other widths fall
    back to the generic GEMM path with a materialised hidden tensor."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, mode, dd, resid):
        ctx.param_refs = (w1, b1, w2, b2)
        x, w1, b1, w2, b2 = _f32c(x), _f32c(w1), _f32c(b1), _f32c(w2), _f32c(b2)
        N, Cin = x.shape
        K, nout = w1.shape[0], w2.shape[0]
        dev = x.device
        fused = (Cin == 32 and K == 1024)
        h = None if fused else torch.empty((N, K), dtype=torch.float32, device=dev)
        raw = torch.empty((N, nout), dtype=torch.float32, device=dev)
        out = torch.empty((N, 3), dtype=torch.float32, device=dev)
        dd = None if dd is None else _f32c(dd)
        ld_resid = 0
        if resid is not None:
            if resid.stride(1) != 1:
                raise L.GeobiError('head: residual rows must be contiguous')
            ld_resid = resid.stride(0)
        L.call('geobi_head_fwd', L.ptr(x), Cin, N, L.ptr(w1), L.ptr(b1), K, L.ptr(w2), L.ptr(b2), nout, LEAK, mode,
               L.ptr(dd), None if resid is None else resid.data_ptr(), ld_resid, L.ptr(h), L.ptr(raw), L.ptr(out),
               L.stream())
        ctx.mode = mode
        ctx.has_dd, ctx.has_h = dd is not None, h is not None
        ctx.save_for_backward(x, w1, b1, w2, raw, dd if dd is not None else raw, h if h is not None else raw)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, w1, b1, w2, raw, dd, h = ctx.saved_tensors
        if not ctx.has_dd:
            dd = None
        if not ctx.has_h:
            h = None
        gout = _f32c(gout)
        N, Cin = x.shape
        K, nout = w1.shape[0], w2.shape[0]
        dev = x.device
        dx = torch.empty_like(x) if (ctx.needs_input_grad[0] or h is None) else None
        direct = [_direct_grad(p_) for p_ in ctx.param_refs]
        if all(d is not None for d in direct):
            dw1, db1, dw2, db2 = direct
            ret = (None, None, None, None)
        else:
            dw1, db1 = torch.empty_like(w1), torch.empty(K, dtype=torch.float32, device=dev)
            dw2, db2 = torch.empty_like(w2), torch.empty(nout, dtype=torch.float32, device=dev)
            ret = (dw1, db1, dw2, db2)
        ws = L.workspace(L.size_query('geobi_head_bwd_ws_bytes', N, Cin, K), dev)
        L.call('geobi_head_bwd', L.ptr(x), Cin, N, L.ptr(w1), L.ptr(b1), K, L.ptr(w2), nout, LEAK, ctx.mode,
               L.ptr(dd), L.ptr(h), L.ptr(raw), L.ptr(gout), L.ptr(dx), L.ptr(dw1), L.ptr(db1), L.ptr(dw2),
               L.ptr(db2), 1 if ret[0] is None else 0, L.ptr(ws), ws.numel(), L.stream())
        return (dx,) + ret + (None, None, None)


# ------------------------------------------------------------------------ losses / metrics
class RowLossFn(Function):
    """scale * sum_i w_i * term(a_i, b_i) over [n, 3] rows (network.py:364-413; kinds in geobi_hip.h)."""

    @staticmethod
    def forward(ctx, a, b, w, kind, scale):
        L.require_device(a, 'prediction')
        a, b = _f32c(a), _f32c(b.detach())
        if a.dim() != 2 or a.shape[1] != 3 or b.shape != a.shape:
            raise L.GeobiError('row loss expects two [n, 3] tensors, got %s and %s' % (tuple(a.shape), tuple(b.shape)))
        w = None if w is None else _f32c(w)
        n = a.shape[0]
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        ws = L.workspace(L.size_query('geobi_row_loss_ws_bytes', n), a.device)
        L.call('geobi_row_loss_fwd', L.ptr(a), L.ptr(b), L.ptr(w), n, int(kind), float(scale), L.ptr(out), L.ptr(ws),
               ws.numel(), L.stream())
        ctx.kind, ctx.scale, ctx.has_w = int(kind), float(scale), w is not None
        ctx.save_for_backward(a, b, w if w is not None else a)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        a, b, w = ctx.saved_tensors
        if ctx.kind > 1:
            raise L.GeobiError('error_v / error_n are metrics: no gradient is defined (the reference never needs one)')
        ga = torch.empty_like(a)
        g = gout.reshape(1).float().contiguous()
        L.call('geobi_row_loss_bwd', L.ptr(a), L.ptr(b), L.ptr(w) if ctx.has_w else None, L.ptr(g), a.shape[0],
               ctx.kind, ctx.scale, L.ptr(ga), L.stream())
        return ga, None, None, None, None


def row_loss(a, b, kind, weights=None, scale=None):
    n = a.shape[0]
    return RowLossFn.apply(a, b, weights, kind, (1.0 / n) if scale is None else scale)
