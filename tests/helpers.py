"""Shared helpers for the parity tests (oracle side is test infrastructure)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def fixture_dual_data(fx, data_cls, device='cpu'):
    """Rebuild (data_v, data_f) from a dualgnn_*.npz fixture with the given Data class."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    dv = data_cls(t(fx['v_x']), t(fx['v_edge_index']).long(), edge_weight=t(fx['v_edge_weight']), y=t(fx['v_y']))
    dv.depth_direction = t(fx['v_depth_direction']) if 'v_depth_direction' in fx else None
    df = data_cls(t(fx['f_x']), t(fx['f_edge_index']).long(), edge_weight=t(fx['f_edge_weight']), y=t(fx['f_y']),
                  fv_indices=t(fx['fv_indices']).long())
    return dv, df


def fixture_clusters(fx, device='cpu'):
    return [torch.from_numpy(fx['cluster_%d' % i]).long().to(device) for i in range(8)]


class ClusterReplay(object):
    """Stands in for graclus: returns a recorded raw cluster vector per call."""

    def __init__(self, clusters):
        self.clusters = list(clusters)
        self.i = 0

    def __call__(self, edge_index, weight=None, num_nodes=None):
        c = self.clusters[self.i]
        self.i += 1
        return c


def install_replay(net, clusters):
    """Order of the 8 graclus calls: gnn_v pooling1 (2), pooling2 (2), gnn_f pooling1 (2), pooling2 (2)."""
    mods = [net.gnn_v.pooling1, net.gnn_v.pooling2, net.gnn_f.pooling1, net.gnn_f.pooling2]
    for k, m in enumerate(mods):
        m.graclus_fn = ClusterReplay(clusters[2 * k:2 * k + 2])


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]
