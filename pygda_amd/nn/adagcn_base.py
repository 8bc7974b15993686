"""``AdaGCNBase`` (pygda/nn/adagcn_base.py:11-181): GCNConv (or PPMIConv) stack with act +
``Dropout(0.1)`` between layers and a linear classifier.  As in the reference, the
``dropout`` given to ``AdaGCNBase`` never reaches the stack (:145 builds ``GNN`` without it,
so the helper's default 0.1 at :39 always applies)."""
import torch
import torch.nn.functional as F
from torch import nn

from .a2gnn_base import global_mean_pool
from .gcn_conv import GCNConv
from .ppmi_conv import PPMIConv
from .linear import DenseLinear


class GNN(nn.Module):
    def __init__(self, in_dim, hid_dim, gnn_type='gcn', num_layers=3, act=F.relu, dropout=0.1, **kwargs):
        super().__init__()
        self.gnn_type, self.act, self.num_layers = gnn_type, act, num_layers
        conv = GCNConv if gnn_type == 'gcn' else PPMIConv
        dims = [in_dim] + [hid_dim] * num_layers
        self.conv_layers = nn.ModuleList(conv(dims[i], dims[i + 1]) for i in range(num_layers))
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, edge_index, batch, mode='node'):
        return self.forward_from(self.conv_layers[0](x, edge_index), edge_index, batch, mode)

    def forward_from(self, h0, edge_index, batch, mode='node', copies=1):
        """The stack continued from the output of its first conv (before activation / dropout).  That output
        is a deterministic function of the inputs and the weights, so the trainer evaluates it once per domain
        and step and shares it between the 11 encoder passes the reference makes per step (10 inside the critic
        loop, 1 for the encoder update): same values, 20 first-layer projections + aggregations fewer."""
        x = h0
        last = len(self.conv_layers) - 1
        for i, conv in enumerate(self.conv_layers):
            if i > 0:
                x = conv(x, edge_index)
            if i == 0 and copies > 1:
                # `copies` passes over the same h0 (the critic loop's re-encodings, under no_grad) as ONE stacked pass:
                # `edge_index` is the block-diagonal graph of the copies; the first activation reads h0 for every copy
                if i < last and self.act is F.relu and x.is_cuda and x.dtype == torch.float32:
                    from ..ops import relu_dropout_copies
                    x = relu_dropout_copies(x, copies, self.dropout.p, self.dropout.training)
                    continue
                x = x.repeat(copies, 1)
            if i < last:
                if self.act is F.relu and x.is_cuda and x.dtype == torch.float32:
                    from ..ops import relu_dropout                  # one kernel each way, no mask tensor
                    x = relu_dropout(x, self.dropout.p, self.dropout.training)
                else:
                    x = self.dropout(self.act(x))
        return global_mean_pool(x, batch) if mode == 'graph' else x


class AdaGCNBase(nn.Module):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=3, dropout=0.1, act=F.relu, gnn_type='gcn',
                 mode='node', **kwargs):
        super().__init__()
        self.encoder = GNN(in_dim=in_dim, hid_dim=hid_dim, gnn_type=gnn_type, act=act, num_layers=num_layers)
        self.cls_model = nn.Sequential(DenseLinear(hid_dim, num_classes))
        self.mode = mode
        self.loss_func = nn.CrossEntropyLoss()

    def forward(self, data):
        batch = None if self.mode == 'node' else data.batch
        return self.encoder(data.x, data.edge_index, batch, mode=self.mode)

    def first_conv(self, data):
        return self.encoder.conv_layers[0](data.x, data.edge_index)

    def forward_from(self, h0, data, copies=1):
        batch = None if self.mode == 'node' else data.batch
        return self.encoder.forward_from(h0, data.edge_index, batch, mode=self.mode, copies=copies)
