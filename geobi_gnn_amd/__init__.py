"""MI355X-native bi-domain mesh-graph convolution engine (hot path of zhangyk18/GeoBi-GNN).

Host-side mirror of the reference's nn.Module surface over a C-ABI HIP library
(include/geobi_hip.h).  Importing the package is cheap; the HIP library is loaded on
first use and its absence is an error, never a CPU fallback.
"""
__version__ = '0.1.0'
