"""``SAGEConv`` and ``GINConv`` with the semantics ``GNNBase`` relies on from PyG
(pygda/nn/gnn_base.py:72-79,88-95), on the MI355X aggregation kernel: both are the plain
sum aggregation ``A x`` (unit weights, no self loops, no normalisation) followed by dense
maps -- SAGE divides by the in-degree (mean) and adds a root projection, GIN adds
``(1 + eps) x`` and applies its MLP."""
import torch
from torch import nn

from ..graph import as_graph
from ..ops import propagate
from .linear import Linear


def _sum_graph(x, edge_index):
    return as_graph(edge_index, x.size(0), None, False, False, False, "col")


class SAGEConv(nn.Module):
    def __init__(self, in_channels, out_channels, aggr="mean", root_weight=True, bias=True, **kwargs):
        super().__init__()
        if aggr != "mean":
            raise NotImplementedError("GNNBase uses SAGEConv's default mean aggregation")
        self.in_channels, self.out_channels, self.root_weight = in_channels, out_channels, root_weight
        # PyG's own Linear with its default initialisers; each draws in its constructor and again in
        # reset_parameters() (lin_l.W, lin_l.b, lin_r.W twice over: the init RNG order gnn_fit2_sage.npz pins)
        self.lin_l = Linear(in_channels, out_channels, bias=bias, weight_initializer=None)
        if root_weight:
            self.lin_r = Linear(in_channels, out_channels, bias=False, weight_initializer=None)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        if self.root_weight:
            self.lin_r.reset_parameters()

    def forward(self, x, edge_index, size=None):
        g = _sum_graph(x, edge_index)
        deg = (g.rowptr[1:] - g.rowptr[:-1]).clamp(min=1).to(x.dtype).unsqueeze(1)
        out = self.lin_l(propagate(x, g, 1) / deg)
        return out + self.lin_r(x) if self.root_weight else out


def _reset(value):
    if hasattr(value, "reset_parameters"):
        value.reset_parameters()
    else:
        for child in value.children() if hasattr(value, "children") else []:
            _reset(child)


class GINConv(nn.Module):
    def __init__(self, nn_module, eps=0.0, train_eps=False, **kwargs):
        super().__init__()
        self.nn = nn_module
        self.initial_eps = eps
        if train_eps:
            self.eps = nn.Parameter(torch.empty(1))
        else:
            self.register_buffer("eps", torch.empty(1))
        self.reset_parameters()

    def reset_parameters(self):
        """PyG ``GINConv.reset_parameters()``: ``reset(nn)`` re-draws every child that can (the torch ``Linear`` inside
        GNNBase's ``Sequential`` is initialised a second time: gnn_fit2_gin.npz pins the stream), eps back to its start."""
        _reset(self.nn)
        self.eps.data.fill_(self.initial_eps)

    def forward(self, x, edge_index, size=None):
        return self.nn(propagate(x, _sum_graph(x, edge_index), 1) + (1 + self.eps) * x)
