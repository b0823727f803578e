"""TEST INFRASTRUCTURE -- platform-stable random weights for fixtures and parity tests.

numpy's PCG64 stream is stable across platforms and versions, so the same state_dict can
be rebuilt on the GPU box without shipping 3.7 MB of weights per fixture.  Distributions
follow the initialisers the modules use (uniform +-1/sqrt(fan_in) for weights, N(0, 0.1)
for FeaStConv c / bias), widened slightly so every branch of the arithmetic is exercised.
"""
import numpy as np
import torch


def make_state_dict(template, seed=0):
    """template: an nn.Module state_dict (only names and shapes are used)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, t in template.items():
        shape = tuple(t.shape)
        if name.endswith('.c') or (name.endswith('.bias') and len(shape) == 1 and 'conv' in name):
            v = rng.normal(0.0, 0.1, size=shape)
        elif len(shape) >= 2:
            bound = 1.0 / np.sqrt(shape[-1])
            v = rng.uniform(-bound, bound, size=shape)
        else:
            v = rng.uniform(-0.05, 0.05, size=shape)
        out[name] = torch.from_numpy(v.astype(np.float32))
    return out


def checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))
