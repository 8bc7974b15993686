"""``GS_reweight`` / ``GCN_reweight`` / ``ReweightGNN`` (pygda/nn/reweight_gnn.py:51-502): message
passing whose edges carry a structure re-weighting factor ``w_e`` mixed in with ``lmda``:
``m_e = ((1 - lmda) + lmda * w_e) * (norm_e *) f(x_j)``, aggregated (mean / add) at ``edge_index[0]``
(``flow='target_to_source'``).

MI355X mapping: the per-edge coefficient, the optional gcn normalisation and the 1/count of the mean
all fold into the VALUES of one CSR (rows = ``edge_index[0]``), rebuilt only when the weights change
(every ``ew_freq`` epochs in StruRW), and -- the per-edge transform being linear -- ``f`` is applied
to the node rows once instead of to every gathered message.  One aggregation launch per layer."""
import torch
import torch.nn.functional as F
from torch import nn

from ..graph import build_csr
from ..ops import propagate
from .linear import Linear, zeros
from .linear import DenseLinear


class _ReweightedGraphCache:
    """CSR of ``coef_e (* norm_e) (/ count(row))`` keyed on the identity and version of the edge
    tensors (held alive by the entry, so an address is never recycled under the key)."""

    def __init__(self):
        self._key, self._graph, self._hold = None, None, None

    def get(self, edge_index, edge_rw, lmda, n, gcn_norm, mean):
        key = (edge_index.data_ptr(), edge_index._version, edge_rw.data_ptr(), edge_rw._version, float(lmda), n,
               gcn_norm, mean)
        if key != self._key:
            row, col = edge_index[0], edge_index[1]
            val = (1.0 - lmda) + lmda * edge_rw.detach().to(torch.float32)
            if gcn_norm:                                  # gcn_norm(ones, no self loops): degree over col
                deg = torch.bincount(col, minlength=n).to(torch.float32)
                dis = deg.pow(-0.5)
                dis.masked_fill_(dis == float("inf"), 0)
                val = val * dis[row] * dis[col]
            if mean:                                      # PyG 'mean': by the NUMBER of messages at the row
                cnt = torch.bincount(row, minlength=n).clamp(min=1).to(torch.float32)
                val = val / cnt[row]
            # messages flow edge_index[1] -> edge_index[0]
            self._graph = build_csr(torch.stack([col, row]), n, val, add_self_loops=False, normalize=False)
            self._key, self._hold = key, (edge_index, edge_rw)
        return self._graph


class GS_reweight(nn.Module):
    def __init__(self, in_channels, out_channels, reducer, normalize_embedding=False):
        super().__init__()
        if reducer not in ("mean", "add"):
            raise NotImplementedError(f"aggregation {reducer!r} is outside the covered StruRW configurations")
        self.aggr = reducer
        self.lin = DenseLinear(in_channels, out_channels)
        self.agg_lin = DenseLinear(out_channels + in_channels, out_channels)
        self.normalize_emb = normalize_embedding
        self._graphs = _ReweightedGraphCache()

    def forward(self, x, edge_index, edge_weight, lmda):
        g = self._graphs.get(edge_index, edge_weight, lmda, x.size(0), False, self.aggr == "mean")
        aggr = propagate(self.lin(x), g, 1)                                   # reweight_gnn.py:308-310 + mean
        out = F.relu(self.agg_lin(torch.cat((aggr, x), dim=-1)))              # :342-347
        return F.normalize(out, p=2, dim=-1) if self.normalize_emb else out


class GCN_reweight(nn.Module):
    def __init__(self, in_channels, out_channels, aggr, improved=False, cached=False, add_self_loops=False,
                 normalize=True, bias=True, **kwargs):
        super().__init__()
        if aggr not in ("mean", "add") or improved or add_self_loops:
            raise NotImplementedError("GCN_reweight: only the configurations ReweightGNN builds are covered")
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.normalize = aggr != "add"                                        # :96-99
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._graphs = _ReweightedGraphCache()

    def reset_parameters(self):
        self.lin.reset_parameters()
        zeros(self.bias)

    def forward(self, x, edge_index, edge_weight, lmda):
        g = self._graphs.get(edge_index, edge_weight, lmda, x.size(0), self.normalize, self.aggr == "mean")
        return propagate(self.lin(x), g, 1, self.bias)                        # :162-167, :222-224


class ReweightGNN(nn.Module):
    def __init__(self, input_dim, gnn_dim, output_dim, cls_dim, gnn_layers=3, cls_layers=2, backbone='GS',
                 pooling='mean', dropout=0.5, bn=False, rw_lmda=1.0, **kwargs):
        super().__init__()
        if backbone == 'GCN':
            self.prop_input = GCN_reweight(input_dim, gnn_dim, pooling)
            self.prop_hidden = GCN_reweight(gnn_dim, gnn_dim, pooling)
        elif backbone == 'GS':
            self.prop_input = GS_reweight(input_dim, gnn_dim, pooling)
            self.prop_hidden = GS_reweight(gnn_dim, gnn_dim, pooling)
        else:
            raise ValueError(f"unknown backbone {backbone!r}")
        self.dropout, self.bn, self.lmda = dropout, bn, rw_lmda
        self.conv = nn.ModuleList([self.prop_input] + [self.prop_hidden] * (gnn_layers - 1))   # :435-438: shared module
        self.bns = nn.ModuleList(nn.BatchNorm1d(gnn_dim) for _ in range(gnn_layers - 1))
        self.bn_mlp = nn.BatchNorm1d(cls_dim)
        dims = [gnn_dim, output_dim] if cls_layers == 1 else [gnn_dim] + [cls_dim] * (cls_layers - 1) + [output_dim]
        self.mlp_classify = nn.ModuleList(DenseLinear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, data, h):
        x, edge_index, edge_weight = h, data.edge_index, data.edge_weight
        for layer in self.conv:
            x = F.relu(layer(x, edge_index, edge_weight, self.lmda))
            x = F.dropout(x, p=self.dropout)              # training=True whatever the mode, as in :488
        y = x
        for i, lin in enumerate(self.mlp_classify):
            y = lin(y)
            if i != len(self.mlp_classify) - 1:
                if self.bn:
                    y = self.bn_mlp(y)
                y = F.relu(y)
        return x, y
