"""``GCNConv`` as the reference uses PyG's (grade_base.py:58-61, adagcn_base.py:49-52,
gnn_base.py:65-71): lin (glorot, no bias) -> destination-degree gcn_norm with self loops
-> one aggregation -> + bias.  Same kernels as :class:`PropGCNConv` with ``prop_nums=1``."""
from .prop_gcn_conv import PropGCNConv


class GCNConv(PropGCNConv):
    def forward(self, x, edge_index, edge_weight=None):
        return super().forward(x, edge_index, 1, edge_weight)
