"""TEST INFRASTRUCTURE (build container only) -- stand-in so the reference orchestration files can be
imported by oracle/gen_golden.py.  NOT the real package: every symbol re-exports the restatement in
oracle/pyg_ops.py, so results pinned through it cover the reference glue only, not these primitives."""
from . import data, nn, utils  # noqa: F401
