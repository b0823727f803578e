"""TEST INFRASTRUCTURE -- CPU restatement of the reference's hot-path orchestration.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates, on top of oracle/pyg_ops.py, the live code of
  /root/reference/code/network.py:254-343   GNNModule, DualGNN
  /root/reference/code/network.py:364-413   loss_v, loss_n, dual_loss, error_v, error_n
  /root/reference/code/net_util.py:56-245   PoolingLayer (all edge_weight_type branches)
  /root/reference/code/net_util.py:289-380  pool_edge, pool_face, pooling, pooling_pre, pooling_run
  /root/reference/code/data_util.py:182-198 computer_face_normal
  /root/reference/code/data_util.py:529-556 update_position2
including the reference's in-place mutations of its inputs.  Parity status: the
orchestration is checked against the reference's own network.py / net_util.py /
data_util.py imported in the build container through oracle/shims (see
oracle/gen_golden.py); the third-party primitives underneath stay PARITY UNPINNED.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import pyg_ops as P

LEAK = 0.2


def computer_face_normal(points, fv_indices):
    # data_util.py:195-197; dim=1 stated explicitly (torch.cross without dim picks the
    # first size-3 axis, which differs only for a mesh of exactly 3 faces)
    tri = points[fv_indices]
    return F.normalize(torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1), dim=1)


# --------------------------------------------------------------------------- pooling
def pool_edge(cluster, edge_index, edge_attr=None, op='mean'):
    n = cluster.size(0)
    ei = cluster[edge_index.view(-1)].view(2, -1)
    ei, edge_attr = P.remove_self_loops(ei, edge_attr)
    if ei.numel() > 0:
        ei, edge_attr = P.coalesce(ei, edge_attr, n, n, op=op)
    return ei, edge_attr


def pool_face(cluster, fv_indices):
    face = cluster[fv_indices.view(-1)].view(-1, 3)
    bad = (face[:, 0] == face[:, 1]) | (face[:, 0] == face[:, 2]) | (face[:, 1] == face[:, 2])
    return face[~bad]


def _feature_gauss(x, edge_index, denom):
    d = x[edge_index]
    return (((d[0] - d[1]) ** 2).sum(1) / (-denom)).exp()


def _minmax(v):
    return (v - v.min()) / (v.max() - v.min() + 1e-12)


class PoolingLayer(nn.Module):
    def __init__(self, in_channel, pool_type='max', pool_step=2, edge_weight_type=0, wei_param=2):
        super().__init__()
        assert pool_type in ('max', 'mean')
        self.pool_type, self.pool_step = pool_type, pool_step
        self.edge_weight_type, self.wei_param = edge_weight_type, wei_param
        if edge_weight_type in (4, 5):
            self.lin = nn.Linear(in_channel, in_channel)
        if edge_weight_type in (3, 4, 5):
            self.att_l = nn.Parameter(torch.empty(1, in_channel))
            self.att_r = nn.Parameter(torch.empty(1, in_channel))
            nn.init.xavier_uniform_(self.att_l.data, gain=1.414)
            nn.init.xavier_uniform_(self.att_r.data, gain=1.414)
        self.unpooling_indices = None
        self.graclus_fn = P.graclus        # tests swap this to replay fixed clusters
        self.last_clusters = None          # raw (pre-relabel) cluster vectors of the last call

    def _attention(self, x, ei):
        al, ar = (x * self.att_l).sum(-1), (x * self.att_r).sum(-1)
        return torch.sigmoid((al[ei[0]] + ar[ei[1]]) + (al[ei[1]] + ar[ei[0]]))

    def _get_edge_weight(self, data):
        w = getattr(data, 'edge_weight', None)
        ei, w = P.remove_self_loops(data.edge_index, w)
        if ei.numel() == 0:
            return None
        data.edge_index, data.edge_weight = ei, w        # net_util.py:166-167 writes back
        t = self.edge_weight_type
        if t == -1:
            return None
        if t == 0:
            return w
        if t == 1:
            return _feature_gauss(data.x, ei, self.wei_param)
        if t == 2:
            return w * _feature_gauss(data.x, ei, self.wei_param)
        if t == 3:
            return self._attention(data.x, ei)
        if t in (4, 5):
            a = self._attention(F.leaky_relu(self.lin(data.x), LEAK), ei)
            return a if t == 4 else (a + w) / 2
        if t == 6:
            return _minmax(w)
        if t == 7:
            d = data.x[ei]
            return _minmax(-((d[0] - d[1]) ** 2).sum(1))
        if t == 8:
            return _minmax(_feature_gauss(data.x, ei, 2))
        if t == 9:
            return _minmax(w) + _minmax(_feature_gauss(data.x, ei, 2))
        if t == 10:
            return w + _feature_gauss(data.x, ei, 2)
        return w

    def forward(self, data, visual=False):
        edge_weight = self._get_edge_weight(data)
        x, edge_index, pos = data.x, data.edge_index, getattr(data, 'pos', None)
        edge_dual = getattr(data, 'edge_dual', None)
        face = getattr(data, 'fv_indices', None)
        clusts, raw = [], []
        for _ in range(self.pool_step):
            cluster = self.graclus_fn(edge_index, edge_weight, x.shape[0])
            raw.append(cluster)
            cluster, _ = P.consecutive_cluster(cluster)
            clusts.append(cluster)
            x = P.scatter(x, cluster, dim=0, reduce=self.pool_type)
            edge_index, edge_weight = pool_edge(cluster, edge_index, edge_weight)
            pos = None if pos is None else P.pool_pos(cluster, pos)
            edge_dual = None if edge_dual is None else cluster[edge_dual]
            if edge_index.numel() == 0:
                break
        clust = clusts[-1]
        for c in clusts[-2::-1]:
            clust = clust[c]
        self.unpooling_indices = clust
        self.last_clusters = raw
        return P.Data(x, edge_index, edge_dual=edge_dual, edge_weight=edge_weight, pos=pos, fv_indices=face)

    def unpooling(self, x):
        return x if self.unpooling_indices is None else x[self.unpooling_indices]


def _compose(clusts):
    clust = clusts[-1]
    for c in clusts[-2::-1]:
        clust = clust[c]
    return clust


def pooling(data, p_type='max', level=2, wei_type=0, graclus_fn=P.graclus):
    x, pos, edge_index = data.x, getattr(data, 'pos', None), data.edge_index
    if wei_type == 0:
        edge_weight = getattr(data, 'edge_weight', None)
    elif wei_type == 1:
        nrm = (x ** 2).sum(1)[edge_index]
        edge_weight = ((nrm[0] - nrm[1]) ** 2 / (-2)).exp()
    else:
        edge_weight = _feature_gauss(x, edge_index, 2)
    clusts = []
    for _ in range(level):
        cluster, _ = P.consecutive_cluster(graclus_fn(edge_index, edge_weight, x.shape[0]))
        clusts.append(cluster)
        x = P.scatter(x, cluster, dim=0, reduce=p_type)
        pos = None if pos is None else P.pool_pos(cluster, pos)
        edge_index, edge_weight = pool_edge(cluster, edge_index, edge_weight)
        if edge_index.numel() == 0:
            break
    return P.Data(x, edge_index, pos=pos, edge_weight=edge_weight), _compose(clusts)


def pooling_pre(data, step=2, level=2, graclus_fn=P.graclus):
    edge_index, edge_weight = data.edge_index, getattr(data, 'edge_weight', None)
    for i in range(1, level + 1):
        clusters = []
        for _ in range(step):
            cluster, _ = P.consecutive_cluster(graclus_fn(edge_index, edge_weight))
            clusters.append(cluster)
            edge_index, edge_weight = pool_edge(cluster, edge_index, edge_weight)
        setattr(data, 'pool_l%d' % i, {'clusters': clusters, 'cluster_inv': _compose(clusters)})
    data.edge_weight = None
    return data


def pooling_run(data, pool_info, p_type='max'):
    x, pos, edge_index = data.x, getattr(data, 'pos', None), data.edge_index
    for clust in pool_info['clusters']:
        x = P.scatter(x, clust, dim=0, reduce=p_type)
        pos = None if pos is None else P.pool_pos(clust, pos)
        edge_index, _ = pool_edge(clust, edge_index)
    return P.Data(x, edge_index, pos=pos)


# --------------------------------------------------------------------------- networks
class GNNModule(nn.Module):
    def __init__(self, in_channel=6, pool_type='max', pool_step=2, edge_weight_type=0, wei_param=2):
        super().__init__()
        C = P.FeaStConv
        self.l_conv1 = C(in_channel, 32, 9)
        self.pooling1 = PoolingLayer(32, pool_type, pool_step, edge_weight_type, wei_param)
        self.l_conv2 = C(32, 64, 9)
        self.pooling2 = PoolingLayer(64, pool_type, pool_step, edge_weight_type, wei_param)
        self.l_conv3 = C(64, 128, 9)
        self.l_conv4 = C(128, 128, 9)
        self.r_conv1 = C(128, 64, 9)
        self.r_conv2 = C(128, 64, 9)
        self.r_conv3 = C(64, 32, 9)
        self.r_conv4 = C(64, 32, 9)

    def forward(self, d1, plot_pool=False):
        act = lambda t: F.leaky_relu(t, LEAK)
        d1.x = act(self.l_conv1(d1.x, d1.edge_index))
        d2 = self.pooling1(d1)
        d2.x = act(self.l_conv2(d2.x, d2.edge_index))
        d3 = self.pooling2(d2)
        d3.x = act(self.l_conv3(d3.x, d3.edge_index))
        d3.x = act(self.l_conv4(d3.x, d3.edge_index))
        up2 = self.r_conv1(self.pooling2.unpooling(d3.x), d2.edge_index)       # no activation
        d2.x = act(self.r_conv2(torch.cat((d2.x, up2), 1), d2.edge_index))
        up1 = self.r_conv3(self.pooling1.unpooling(d2.x), d1.edge_index)       # no activation
        d1.x = torch.cat((d1.x, up1), 1)
        return act(self.r_conv4(d1.x, d1.edge_index))


class DualGNN(nn.Module):
    def __init__(self, force_depth=False, pool_type='max', edge_weight_type=10, wei_param=2):
        super().__init__()
        self.force_depth = force_depth
        self.gnn_v = GNNModule(6, pool_type, 2, edge_weight_type, wei_param)
        self.fc_v1 = nn.Linear(32, 1024)
        self.fc_v2 = nn.Linear(1024, 1 if force_depth else 3)
        self.gnn_f = GNNModule(12, pool_type, 2, edge_weight_type, wei_param)
        self.fc_f1 = nn.Linear(32, 1024)
        self.fc_f2 = nn.Linear(1024, 3)

    def forward(self, dual_data):
        data_v, data_f = dual_data
        xyz = data_v.x[:, :3]
        feat_v = self.fc_v2(F.leaky_relu(self.fc_v1(self.gnn_v(data_v)), LEAK))
        if self.force_depth:
            feat_v = feat_v * data_v.depth_direction
        feat_v = feat_v + xyz
        cent = feat_v[data_f.fv_indices].mean(1)
        nrm = computer_face_normal(feat_v, data_f.fv_indices)
        data_f.x = torch.cat((data_f.x, cent, nrm), 1)
        feat_f = self.fc_f2(F.leaky_relu(self.fc_f1(self.gnn_f(data_f)), LEAK))
        return feat_v, F.normalize(feat_f, dim=1), None


# ----------------------------------------------------------------------------- losses
def loss_v(vp, v, dis='L2', apply_icp=False):
    if dis == 'L1':
        return (vp - v).abs().sum(1).mean()
    if dis == 'L2':
        return (vp - v).pow(2).sum(1).mean()
    raise ValueError('loss_v: %r needs packages the reference never imports (network.py:12-13)' % dis)


def loss_n(np_, n, norm='L1', fc_p=None, fc=None):
    if norm == 'L1':
        return (np_ - n).abs().sum(1).mean()
    if norm == 'L2':
        return (np_ - n).pow(2).sum(1).mean()
    raise ValueError('loss_n: %r needs packages the reference never imports' % norm)


def dual_loss(loss_v, loss_n, v_scale=1, n_scale=1, alpha=None):
    if alpha is None:
        return loss_v * v_scale + loss_n * n_scale
    return alpha * loss_v * v_scale + (1 - alpha) * loss_n * n_scale


def error_v(vp, v):
    return (vp - v).pow(2).sum(1).pow(0.5).mean()


def error_n(np_, n):
    val = torch.clamp(1 - (np_ - n).pow(2).sum(1) / 2, min=-1, max=1)
    return (torch.acos(val) * 180 / math.pi).mean()


def laplacian_loss(vp, v, edge_idx_v, normal=None):
    ei, _ = P.remove_self_loops(edge_idx_v)

    def lap(p):
        out = P.scatter(p[ei[0]] - p[ei[1]], ei[0], dim=0, reduce='mean')
        return out if normal is None else normal * (out * normal).sum(1, keepdim=True)
    return (lap(vp) - lap(v)).abs().sum(1).mean()


# --------------------------------------------------------------- vertex update (row f1)
def update_position2(points, fv_indices, vf_indices, face_normals, n_iter=20, depth_direction=None):
    cnt = torch.clamp((vf_indices > -1).sum(-1, keepdim=True), min=1)
    fn = torch.cat((face_normals, face_normals.new_zeros((1, 3))))
    adj_n = fn[vf_indices]                       # -1 indexes the appended zero row
    for _ in range(n_iter):
        cent = points[fv_indices].mean(1)
        v_cx = cent[vf_indices] - points.unsqueeze(1)
        step = (adj_n * (adj_n * v_cx).sum(-1, keepdim=True)).sum(1) / cnt
        if depth_direction is not None:
            step = (step * depth_direction).sum(1, keepdim=True) * depth_direction
        points = points + step
    return points
