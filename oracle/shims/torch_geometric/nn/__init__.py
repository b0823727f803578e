"""TEST INFRASTRUCTURE (build container only) -- stand-in so the reference orchestration files can be
imported by oracle/gen_golden.py.  NOT the real package: every symbol re-exports the restatement in
oracle/pyg_ops.py, so results pinned through it cover the reference glue only, not these primitives."""
from oracle.pyg_ops import FeaStConv, graclus  # noqa: F401
from . import pool  # noqa: F401


class _Unavailable(object):
    def __init__(self, *a, **k):
        raise NotImplementedError('only FeaStConv is on the hot path (SURVEY.md section 2 rows 11-12)')


GCNConv = GATConv = _Unavailable
