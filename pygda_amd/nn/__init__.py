from .linear import Linear, glorot, zeros
from .prop_gcn_conv import PropGCNConv, gcn_norm
from .gcn_conv import GCNConv
from .cached_gcn_conv import CachedGCNConv
from .ppmi_conv import PPMIConv, ppmi_edges
from .reverse_layer import GradReverse
from .attention import Attention
from .a2gnn_base import A2GNNBase, global_mean_pool
from .grade_base import GRADEBase
from .udagcn_base import UDAGCNBase
from .adagcn_base import AdaGCNBase
from .sage_gin_conv import SAGEConv, GINConv
from .gat_conv import GATConv
from .gnn_base import GNNBase
from .dgsda_base import BernProp, DGSDABase
from .reweight_gnn import GCN_reweight, GS_reweight, ReweightGNN
from .mixup_gcnconv import MixUpGCNConv
from .mixup_base import MixupBase, ShuffledEdges

__all__ = ["Linear", "glorot", "zeros", "PropGCNConv", "gcn_norm", "GCNConv", "CachedGCNConv", "PPMIConv",
           "ppmi_edges", "GradReverse", "Attention", "A2GNNBase", "GRADEBase", "UDAGCNBase", "AdaGCNBase", "SAGEConv", "GINConv", "GATConv", "GNNBase",
           "global_mean_pool", "BernProp", "DGSDABase", "GCN_reweight", "GS_reweight", "ReweightGNN", "MixUpGCNConv", "MixupBase",
           "ShuffledEdges"]
