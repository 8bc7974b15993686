from .linear import Linear, glorot, zeros
from .prop_gcn_conv import PropGCNConv, gcn_norm
from .gcn_conv import GCNConv
from .cached_gcn_conv import CachedGCNConv
from .reverse_layer import GradReverse
from .attention import Attention
from .a2gnn_base import A2GNNBase, global_mean_pool
from .grade_base import GRADEBase

__all__ = ["Linear", "glorot", "zeros", "PropGCNConv", "gcn_norm", "GCNConv", "CachedGCNConv",
           "GradReverse", "Attention", "A2GNNBase", "GRADEBase", "global_mean_pool"]
