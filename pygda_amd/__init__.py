"""pygda_amd -- MI355X-native graph-domain-adaptation training path behind pygda's own
``BaseGDA.fit()/predict()`` and ``pygda.nn`` operator names.  See DESIGN.md."""
from . import _lib
from ._cpu import respect_cpu_quota

respect_cpu_quota()          # the intra-op pool within the container's CPU quota (pygda_amd/_cpu.py: why)
from .data import Data, NeighborLoader, to_undirected
from .graph import CSRGraph, build_csr, graph_cache
from . import nn, utils, metrics, models, datasets, ops

__version__ = "0.1.0"
__all__ = ["Data", "NeighborLoader", "to_undirected", "CSRGraph", "build_csr", "graph_cache",
           "nn", "utils", "metrics", "models", "datasets", "ops"]
