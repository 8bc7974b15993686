"""``GATConv(heads=1, concat=False)`` as ``GNNBase(gnn='gat')`` uses it
(pygda/nn/gnn_base.py:80-87).  The edge-softmax attention aggregation needs its own fused
kernel (segmented max / sum per destination + weighted gather); it is the next kernel on
the list ('What comes next') and is not built yet."""
from torch import nn


class GATConv(nn.Module):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, **kwargs):
        super().__init__()
        raise NotImplementedError(
            "GATConv: the edge-softmax aggregation kernel is not built yet (DESIGN.md §8 'next'); "
            "use gnn in ('gcn', 'sage', 'gin')")
