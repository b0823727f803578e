"""FeaStConv module with the state-dict layout of torch_geometric >= 2.0.

Constructed 16x by the reference (/root/reference/code/network.py:258-268 ``FeaStConv(in, out, 9)``).
Parameters: ``lin.weight [heads*out, in]``, ``u.weight [heads, in]``, ``c [heads]``, ``bias [out]``.
PyG 1.x checkpoints (``weight [in, heads*out]``, ``u [in, heads]``) are accepted on load by
transposing.  The arithmetic runs in libgeobi_hip.so (geobi_feast_fwd / geobi_feast_bwd).
"""
import math

import torch
from torch import nn

from . import ops
from .graph import Graph, graph_of


class _BareLinear(nn.Module):
    """Holds a bias-free ``weight`` so the keys read ``lin.weight`` / ``u.weight``."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))


class FeaStConv(nn.Module):
    def __init__(self, in_channels, out_channels, heads=1, add_self_loops=True, bias=True):
        super().__init__()
        if heads != 9:
            raise NotImplementedError('the HIP path is built for heads=9 (every FeaStConv of the reference net)')
        if not add_self_loops or not bias:
            raise NotImplementedError('the reference net always uses add_self_loops=True, bias=True')
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.lin = _BareLinear(in_channels, heads * out_channels)
        self.u = _BareLinear(in_channels, heads)
        self.c = nn.Parameter(torch.empty(heads))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels)      # PyG Linear(weight_initializer='uniform')
        nn.init.uniform_(self.lin.weight, -bound, bound)
        nn.init.uniform_(self.u.weight, -bound, bound)
        nn.init.normal_(self.c, mean=0.0, std=0.1)
        nn.init.normal_(self.bias, mean=0.0, std=0.1)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # PyG 1.x naming: weight [in, heads*out], u [in, heads]
        for old, new in (('weight', 'lin.weight'), ('u', 'u.weight')):
            if prefix + old in state_dict and prefix + new not in state_dict:
                state_dict[prefix + new] = state_dict.pop(prefix + old).t().contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x, edge_index, x2=None, slope=1.0):
        """x [N, in] (or x | x2 halves), edge_index [2, E] int64 (row = source, col = target) or the
        level's cached ``Graph`` (what GNNModule passes, so pooled levels never build a COO tensor).

        ``slope`` fuses the leaky_relu that follows most layers; ``x2`` fuses the skip
        concatenation ``cat((x, x2), 1)`` (network.py:292,298).
        """
        g = edge_index if isinstance(edge_index, Graph) else graph_of(edge_index, x.shape[0])
        return ops.feast_conv(x, g, self.lin.weight, self.u.weight, self.c, self.bias, slope=slope, x2=x2)

    def __repr__(self):
        return 'FeaStConv(%d, %d, heads=%d)' % (self.in_channels, self.out_channels, self.heads)
