"""``UDAGCNBase`` (pygda/nn/udagcn_base.py:9-267): a GCN view (``CachedGCNConv`` stack) and a
PPMI view (``PPMIConv`` stack sharing the SAME weight/bias Parameters) fused by view
attention; linear classifier; MLP domain discriminator.

Two behaviours of the reference are kept on purpose, because they change its outputs:
the per-layer ``Dropout(0.1)`` modules live in a plain Python list (:47) -- not registered, so
never switched to eval and active even in ``predict()`` -- and the constructor's ``dropout``
argument is ignored."""
import torch
import torch.nn.functional as F
from torch import nn

from .attention import Attention
from .cached_gcn_conv import CachedGCNConv
from .ppmi_conv import PPMIConv
from .linear import DenseLinear


class GNN(nn.Module):
    def __init__(self, in_dim, hid_dim, gnn_type='gcn', num_layers=3, base_model=None, act=F.relu, **kwargs):
        super().__init__()
        shared = None if base_model is None else list(base_model.conv_layers)
        self.dropout_layers = [nn.Dropout(0.1) for _ in range(num_layers)]      # plain list: see header
        self.gnn_type, self.act = gnn_type, act
        conv = PPMIConv if gnn_type == 'ppmi' else CachedGCNConv
        dims = [in_dim] + [hid_dim] * num_layers
        self.conv_layers = nn.ModuleList(
            conv(dims[i], dims[i + 1], weight=None if shared is None else shared[i].weight,
                 bias=None if shared is None else shared[i].bias, **kwargs) for i in range(num_layers))

    def forward(self, x, edge_index, cache_name):
        last = len(self.conv_layers) - 1
        for i, conv in enumerate(self.conv_layers):
            x = conv(x, edge_index, cache_name)
            if i < last:
                drop = self.dropout_layers[i]
                if self.act is F.relu and x.is_cuda and x.dtype == torch.float32:
                    from ..ops import relu_dropout                  # one kernel each way, no mask tensor
                    x = relu_dropout(x, drop.p, drop.training)
                else:
                    x = drop(self.act(x))
        return x


class UDAGCNBase(nn.Module):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=3, dropout=0.1, act=F.relu, ppmi=True,
                 adv_dim=40, **kwargs):
        super().__init__()
        self.ppmi = ppmi
        self.encoder = GNN(in_dim=in_dim, hid_dim=hid_dim, gnn_type='gcn', act=act, num_layers=num_layers)
        if ppmi:
            self.ppmi_encoder = GNN(in_dim=in_dim, hid_dim=hid_dim, base_model=self.encoder,
                                    num_layers=num_layers, gnn_type='ppmi', path_len=10)
        self.cls_model = nn.Sequential(DenseLinear(hid_dim, num_classes))
        self.domain_model = nn.Sequential(DenseLinear(hid_dim, adv_dim), nn.ReLU(), nn.Dropout(0.1),
                                          DenseLinear(adv_dim, 2))
        self.att_model = Attention(hid_dim)
        self.models = [self.encoder, self.cls_model, self.domain_model]
        if ppmi:
            self.models.extend([self.ppmi_encoder, self.att_model])
        self.loss_func = nn.CrossEntropyLoss()

    def gcn_encode(self, data, cache_name, mask=None):
        out = self.encoder(data.x, data.edge_index, cache_name)
        return out if mask is None else out[mask]

    def ppmi_encode(self, data, cache_name, mask=None):
        out = self.ppmi_encoder(data.x, data.edge_index, cache_name)
        return out if mask is None else out[mask]

    def encode(self, data, cache_name, mask=None):
        gcn_output = self.gcn_encode(data, cache_name, mask)
        if not self.ppmi:
            return gcn_output
        return self.att_model([gcn_output, self.ppmi_encode(data, cache_name, mask)])
