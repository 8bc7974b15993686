"""``GRADEBase`` (pygda/nn/grade_base.py:9-202): L GCNConv layers whose every output,
plus the logits, are concatenated into the feature the domain loss sees."""
import torch
import torch.nn.functional as F
from torch import nn

from .a2gnn_base import global_mean_pool
from .gcn_conv import GCNConv
from .linear import DenseLinear


class GRADEBase(nn.Module):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=1, dropout=0.1, act=F.relu,
                 disc="JS", mode="node", **kwargs):
        super().__init__()
        self.in_dim, self.hid_dim, self.num_classes = in_dim, hid_dim, num_classes
        self.num_layers, self.dropout, self.act, self.mode = num_layers, dropout, act, mode
        widths = [in_dim] + [hid_dim] * num_layers
        self.convs = nn.ModuleList(GCNConv(a, b) for a, b in zip(widths[:-1], widths[1:]))
        self.cls = DenseLinear(hid_dim, num_classes)
        feat_width = hid_dim * num_layers + num_classes * (1 if disc == "JS" else 2)
        self.discriminator = nn.Sequential(DenseLinear(feat_width, 2))
        self.criterion = nn.CrossEntropyLoss()

    def forward(self, data):
        batch = None if self.mode == "node" else data.batch
        x, feats = self.feat_bottleneck(data.x, data.edge_index, batch)
        x = self.feat_classifier(x)
        feats.append(x)
        return x, torch.cat(feats, dim=1)

    def feat_bottleneck(self, x, edge_index, batch):
        feats = []
        fused = self.act is F.relu and x.is_cuda and x.dtype == torch.float32
        for conv in self.convs:
            if fused:                    # activation + dropout as one kernel each way (no mask tensor)
                from ..ops import relu_dropout
                x = relu_dropout(conv(x, edge_index), self.dropout, self.training)
            else:
                x = F.dropout(self.act(conv(x, edge_index)), p=self.dropout, training=self.training)
            feats.append(x if self.mode == "node" else global_mean_pool(x, batch))
        if self.mode == "graph":
            x = global_mean_pool(x, batch)
        return x, feats

    def feat_classifier(self, x):
        return self.cls(x)

    def one_hot_embedding(self, labels):
        return torch.eye(self.num_classes, device=labels.device)[labels]
