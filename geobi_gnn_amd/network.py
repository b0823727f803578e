"""DualGNN / GNNModule and the loss + metric functions behind the reference's ``network`` surface.

Mirrors /root/reference/code/network.py:254-343 (GNNModule, DualGNN) and :347-413 (losses, errors):
same class names, constructor arguments, child-module names and therefore state-dict keys
(``gnn_v.l_conv1.lin.weight`` ... ``fc_f2.bias``).  The per-layer glue that the reference leaves to
PyTorch eager ops is folded into the HIP calls: the leaky_relu after a conv and the skip
concatenations ride inside geobi_feast_fwd, the heads + residual / normalisation inside
geobi_head_fwd, the centroid/normal coupling inside geobi_face_geom_fwd.
"""
import math

import os

import torch
from torch import nn

from . import _lib as L
from . import executor, ops
from .feast_conv import FeaStConv
from .net_util import PoolingLayer

LEAK = 0.2


class GNNModule(nn.Module):
    def __init__(self, in_channel=6, pool_type='max', pool_step=2, edge_weight_type=0, wei_param=2):
        super().__init__()
        self.l_conv1 = FeaStConv(in_channel, 32, 9)
        self.pooling1 = PoolingLayer(32, pool_type, pool_step, edge_weight_type, wei_param)
        self.l_conv2 = FeaStConv(32, 64, 9)
        self.pooling2 = PoolingLayer(64, pool_type, pool_step, edge_weight_type, wei_param)
        self.l_conv3 = FeaStConv(64, 128, 9)
        self.l_conv4 = FeaStConv(128, 128, 9)

        self.r_conv1 = FeaStConv(128, 64, 9)
        self.r_conv2 = FeaStConv(128, 64, 9)
        self.r_conv3 = FeaStConv(64, 32, 9)
        self.r_conv4 = FeaStConv(64, 32, 9)

    def forward(self, data_r1, plot_pool=False):
        # every level's adjacency is the cached CSR; pooled levels never materialise a COO tensor
        g1 = data_r1.graph(data_r1.x.shape[0])
        # level 0
        data_r1.x = self.l_conv1(data_r1.x, g1, slope=LEAK)
        data_r2 = self.pooling1(data_r1)
        g2 = data_r2.graph()
        # level 1
        data_r2.x = self.l_conv2(data_r2.x, g2, slope=LEAK)
        data_r3 = self.pooling2(data_r2)
        g3 = data_r3.graph()
        # level 2
        data_r3.x = self.l_conv3(data_r3.x, g3, slope=LEAK)
        data_r3.x = self.l_conv4(data_r3.x, g3, slope=LEAK)
        # up to level 1: r_conv1 has no activation; the skip cat feeds r_conv2 as two halves
        up2 = self.r_conv1(self.pooling2.unpooling(data_r3.x), g2)
        data_r2.x = self.r_conv2(data_r2.x, g2, x2=up2, slope=LEAK)
        # up to level 0
        up1 = self.r_conv3(self.pooling1.unpooling(data_r2.x), g1)
        return self.r_conv4(data_r1.x, g1, x2=up1, slope=LEAK)


class _CornerIndex(object):
    """vertex -> corner inverse lists of a face table (FaceGeomFn.backward sums corner gradients through them);
    built on first use, so inference never pays for the sort."""

    def __init__(self, fv32, num_vertices):
        self.fv32, self.num_vertices, self.index = fv32, num_vertices, None

    def get(self):
        if self.index is None:
            self.index = ops.SegmentIndex(self.fv32.view(-1), self.num_vertices)
        return self.index


def mark_face_table(fv, fv32, num_vertices):
    """Attach the validated int32 form of a face table to it (producers that know their ids are in range -- device
    preprocessing, unions of validated tables -- call this so the forward needs no range check, i.e. no host read)."""
    fv._geobi_fv = (fv32, _CornerIndex(fv32, num_vertices), num_vertices)


def _fv_index(data_f, num_vertices):
    """int32 face->vertex table and the (lazy) vertex->corner inverse lists, cached on the tensor."""
    fv = data_f.fv_indices
    cache = getattr(fv, '_geobi_fv', None)
    if cache is None or cache[2] != num_vertices:
        fv32 = fv.to(torch.int32).contiguous()
        if fv32.numel():        # once per mesh (cached): a bad vertex id would be a faulting gather later
            lo, hi = torch.aminmax(fv32)
            if int(lo) < 0 or int(hi) >= num_vertices:
                raise L.GeobiError('fv_indices index vertices outside [0, %d)' % num_vertices)
        mark_face_table(fv, fv32, num_vertices)
        cache = fv._geobi_fv
    return cache[0], cache[1]


class DualGNN(nn.Module):
    def __init__(self, force_depth=False, pool_type='max', edge_weight_type=10, wei_param=2):
        super().__init__()
        self.force_depth = force_depth

        # graph-v
        self.gnn_v = GNNModule(6, pool_type, 2, edge_weight_type, wei_param)
        self.fc_v1 = nn.Linear(32, 1024)
        self.fc_v2 = nn.Linear(1024, 1) if self.force_depth else nn.Linear(1024, 3)

        # graph-f
        self.gnn_f = GNNModule(12, pool_type, 2, edge_weight_type, wei_param)
        self.fc_f1 = nn.Linear(32, 1024)
        self.fc_f2 = nn.Linear(1024, 3)

    def forward(self, dual_data):
        """(data_v, data_f) -> (verts [V,3], unit normals [F,3], None)   (network.py:318-343).

        The whole network is one autograd node (DualGNNFn): the ~70 differentiable ops inside run
        through ops.Tape, which replays their backward passes in reverse order -- same kernels, no
        per-op autograd overhead."""
        data_v, data_f = dual_data
        L.require_device(data_v.x, 'data_v.x')
        params = [p for p in self.parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            verts, normals = DualGNNFn.apply(self, data_v, data_f, *params)
        else:
            with torch.no_grad():
                fast = executor.forward(self, data_v, data_f)       # the whole pass as one library call
                if fast is not None:
                    verts, normals = fast
                else:
                    verts, normals, _ = self._forward_impl(data_v, data_f, ops.Tape(record=False))
        return verts, normals, None

    def _forward_impl(self, data_v, data_f, tape):
        x_v0 = data_v.x.contiguous()          # xyz = x[:, :3] is read in place as the head's residual
        dd = data_v.depth_direction if self.force_depth else None
        with ops.use_tape(tape):
            feat_v = self.gnn_v(data_v)
            verts = ops.apply_op(ops.HeadFn, feat_v, self.fc_v1.weight, self.fc_v1.bias, self.fc_v2.weight,
                                 self.fc_v2.bias, 0, dd, x_v0)
            # new node feature of the facet graph: centroid + normal of the PREDICTED geometry
            fv32, corner_index = _fv_index(data_f, verts.shape[0])
            data_f.x = ops.apply_op(ops.FaceGeomFn, verts, data_f.x, fv32, corner_index)
            feat_f = self.gnn_f(data_f)
            normals = ops.apply_op(ops.HeadFn, feat_f, self.fc_f1.weight, self.fc_f1.bias, self.fc_f2.weight,
                                   self.fc_f2.bias, 1, None, None)
        return verts, normals, tape


# The weight-gradient GEMM of a layer ([x | 1]^T r') depends on the LAST kernel of that layer's backward, so it only
# overlaps with anything if the side stream is joined once per backward of the whole network (the buffers it reads
# are parked in ops._SIDE_KEEP until then) instead of once per layer.  GEOBI_DEFER_JOIN=0 restores per-layer joins.
_DEFER_JOIN = os.environ.get('GEOBI_DEFER_JOIN', '1') == '1'


class DualGNNFn(torch.autograd.Function):
    """DualGNN as a single autograd node.  Inputs after the three leading objects are the module's
    parameters (so autograd routes their gradients); the backward replays the op tape."""

    @staticmethod
    def forward(ctx, net, data_v, data_f, *params):
        ctx.recorded = None
        fast = executor.forward_train(net, data_v, data_f)        # forward + record in one library call
        if fast is not None:
            verts, normals, ctx.recorded = fast
            ctx.params = params
            return verts, normals
        tape = ops.Tape(record=True)
        verts, normals, _ = net._forward_impl(data_v, data_f, tape)
        ctx.tape, ctx.verts, ctx.normals = tape, verts, normals
        ctx.param_ids = [id(p) for p in params]
        return verts, normals

    @staticmethod
    def backward(ctx, g_verts, g_normals):
        if ctx.recorded is not None:
            rec, ctx.recorded = ctx.recorded, None
            grads = executor.backward(rec, g_verts, g_normals, ctx.params)
            ctx.params = None
            # the record (arena, host-side tape) goes when `rec` does; the backward's kernels are already enqueued and
            # the caching allocator reuses the block in stream order
            return (None, None, None) + tuple(grads)
        seeds = {}
        if g_verts is not None:
            seeds[id(ctx.verts)] = g_verts.contiguous()
        if g_normals is not None:
            seeds[id(ctx.normals)] = g_normals.contiguous()
        with ops.deferred_side_join(enable=_DEFER_JOIN):
            leaf = ctx.tape.backward(seeds, clear=False)
        ctx.tape.nodes = []
        ctx.tape = ctx.verts = ctx.normals = None
        return (None, None, None) + tuple(leaf.get(k) for k in ctx.param_ids)


# ---------------------------------------------------------------------------------------
_KIND = {'L1': 0, 'L2': 1}


def loss_v(vp, v, dis='L2', apply_icp=False):
    """network.py:364-377: mean over vertices of sum_c |d| ('L1') or sum_c d^2 ('L2')."""
    if apply_icp:
        raise NotImplementedError('ICP alignment needs pytorch3d, which the reference treats as optional')
    if dis not in _KIND:
        raise NotImplementedError("loss_v: %r relies on kaolin, which the reference never imports" % (dis,))
    return ops.row_loss(vp, v, _KIND[dis])


def loss_n(np, n, norm='L1', fc_p=None, fc=None):
    """network.py:380-389."""
    if norm not in _KIND:
        raise NotImplementedError("loss_n: %r relies on kaolin, which the reference never imports" % (norm,))
    return ops.row_loss(np, n, _KIND[norm])


def dual_loss(loss_v, loss_n, v_scale=1, n_scale=1, alpha=None):
    if alpha is None:
        if v_scale == 1 and n_scale == 1:        # the default: one add, not two multiplications by 1 and an add
            return loss_v + loss_n
        return loss_v * v_scale + loss_n * n_scale
    return alpha * loss_v * v_scale + (1 - alpha) * loss_n * n_scale


def error_v(vp, v):
    """Mean Euclidean distance (network.py:399-404)."""
    return ops.row_loss(vp.detach(), v, 2)


def error_n(np, n):
    """Mean angle in degrees between unit normals (network.py:407-413)."""
    return ops.row_loss(np.detach(), n, 3)


def laplacian_loss(vp, v, edge_idx_v, normal=None):
    keep = edge_idx_v[0] != edge_idx_v[1]
    row, col = edge_idx_v[0][keep], edge_idx_v[1][keep]
    n = vp.shape[0]
    cnt = torch.bincount(row, minlength=n).clamp(min=1).to(vp.dtype).unsqueeze(1)

    def lap(p):
        out = torch.zeros_like(p).index_add_(0, row, p[row] - p[col]) / cnt
        return out if normal is None else normal * (out * normal).sum(1, keepdim=True)
    return (lap(vp) - lap(v)).abs().sum(1).mean()
