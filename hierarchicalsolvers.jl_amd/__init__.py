"""MI355X-native nested-dissection elimination (HierarchicalSolvers.jl hot path).

Host-side mirror of the reference's public API (``src/HierarchicalSolvers.jl:20-28``)
above the C ABI in ``include/hs_solver.h``.  The directory name contains a dot,
so import it through ``hsamd`` (repo root) which registers it as
``hierarchicalsolvers_jl_amd``.
"""
from .nesteddissection import (  # noqa: F401
    NDNode, isleaf, isbranch, depth, symfact, postorder, postorder_nodes, permuted, invperm,
    contigious, parse_elimtree, serialize_elimtree, getinterior, getboundary, flatten_tree, native_symbolic, native_graph_symbolic,
)
from . import problems  # noqa: F401
from .solver import SolverOptions, chkopts, factor, factorize, FactorNode, ldiv, maxrank, trim  # noqa: F401,E402
from ._lib import DimensionMismatch, SingularException, DeviceError, UnsupportedError  # noqa: F401,E402
from . import _lib  # noqa: F401,E402
from . import dist  # noqa: F401,E402
from .gmres import gmres  # noqa: F401,E402
from . import hss  # noqa: F401,E402
