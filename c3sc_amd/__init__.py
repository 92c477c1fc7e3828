"""c3sc_amd -- MI355X-native Bellman-backup hot path of goroda/c3sc.

Scope: the per-fiber value-iteration backup of src/bellman.c +
src/valuefunc.c + src/nodeutil.c, as hand-written gfx950 HIP kernels behind a C-ABI
(include/c3sc_hip.h).  `engine` binds that library; `workloads` holds the benchmark problem
definitions.  There is no CPU fallback: importing `engine` objects without the built library raises.
"""
from . import workloads  # noqa: F401

__all__ = ["workloads"]
