"""``GNNBase`` (pygda/nn/gnn_base.py:11-203): a stack of gcn / sage / gat / gin layers
(act + dropout between them) with a same-type conv classifier in node mode; returns
``log_softmax``."""
import torch.nn.functional as F
from torch import nn

from .a2gnn_base import global_mean_pool
from .gcn_conv import GCNConv
from .sage_gin_conv import GINConv, SAGEConv
from .linear import DenseLinear


def _make(gnn, a, b):
    if gnn == 'gcn':
        return GCNConv(a, b)
    if gnn == 'sage':
        return SAGEConv(a, b)
    if gnn == 'gin':
        return GINConv(nn.Sequential(DenseLinear(a, b)), train_eps=True)
    from .gat_conv import GATConv
    return GATConv(a, b, heads=1, concat=False)


class GNNBase(nn.Module):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=1, dropout=0.1, act=F.relu, gnn='gcn',
                 mode='node', **kwargs):
        super().__init__()
        assert gnn in ('gcn', 'sage', 'gat', 'gin'), 'Invalid gnn backbone'
        self.in_dim, self.hid_dim, self.num_classes = in_dim, hid_dim, num_classes
        self.num_layers, self.dropout, self.gnn, self.act, self.mode = num_layers, dropout, gnn, act, mode
        dims = [in_dim] + [hid_dim] * num_layers
        self.convs = nn.ModuleList(_make(gnn, dims[i], dims[i + 1]) for i in range(num_layers))
        self.cls = _make(gnn, hid_dim, num_classes) if mode == 'node' else DenseLinear(hid_dim, num_classes)

    def forward(self, x, edge_index, edge_weight=None, batch=None):
        x = self.feat_bottleneck(x, edge_index, edge_weight)
        if self.mode == 'graph':
            x = global_mean_pool(x, batch)
        return F.log_softmax(self.feat_classifier(x, edge_index, edge_weight), dim=1)

    def feat_bottleneck(self, x, edge_index, edge_weight=None):
        last = len(self.convs) - 1
        for i, conv in enumerate(self.convs):
            x = conv(x, edge_index, edge_weight)
            if i < last:
                x = F.dropout(self.act(x), p=self.dropout, training=self.training)
        return x

    def feat_classifier(self, x, edge_index, edge_weight=None):
        return self.cls(x, edge_index, edge_weight) if self.mode == 'node' else self.cls(x)
