"""``GATConv(heads=1, concat=False)`` as ``GNNBase(gnn='gat')`` uses it
(pygda/nn/gnn_base.py:80-87), on the fused edge-softmax aggregation kernels
(csrc/gda_gat.hip): ``h = x W^T``; self loops (existing removed, one added per node);
``e_ij = LeakyReLU_0.2(att_src.h_j + att_dst.h_i)``; ``alpha = softmax_j``;
``out_i = sum_j alpha_ij h_j + bias``.  Attention dropout is 0 (the PyG default GNNBase keeps)."""
import weakref

import torch
from torch import nn

from .. import _lib
from ..graph import build_csr
from .linear import Linear, glorot, zeros

_graphs = {}      # attention graphs (self loops, unit weights, edge map), keyed like the normalised ones


def _gat_graph(edge_index, num_nodes):
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes))
    hit = _graphs.get(key)
    if hit is not None and hit[0]() is edge_index:
        return hit[1]
    g = build_csr(edge_index, num_nodes, None, False, True, False, "col", with_edge_map=True)
    _graphs[key] = (weakref.ref(edge_index), g)
    if len(_graphs) > 32:
        for k in [k for k, v in _graphs.items() if v[0]() is None]:
            del _graphs[k]
    return g


class _GatAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, a_src, a_dst, graph, slope):
        h, a_src, a_dst = h.contiguous(), a_src.contiguous(), a_dst.contiguous()
        n, d = h.shape
        out = torch.empty_like(h)
        alpha = torch.empty(graph.nnz_cap, dtype=torch.float32, device=h.device)
        L = _lib.lib()
        _lib.check(L.gda_gat_fwd_f32(_lib.ptr(graph.rowptr), _lib.ptr(graph.colidx), n, d, _lib.ptr(h),
                                     _lib.ptr(a_src), _lib.ptr(a_dst), float(slope), _lib.ptr(out),
                                     _lib.ptr(alpha), _lib.stream()), "gda_gat_fwd_f32")
        ctx.save_for_backward(h, a_src, a_dst, alpha)
        ctx.graph, ctx.slope = graph, float(slope)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, a_src, a_dst, alpha = ctx.saved_tensors
        g = ctx.graph
        n, d = h.shape
        gout = gout.contiguous()
        gh = torch.empty_like(h)
        ga_src, ga_dst = torch.empty_like(a_src), torch.empty_like(a_dst)
        dpre = torch.empty_like(alpha)
        L = _lib.lib()
        _lib.check(L.gda_gat_bwd_f32(_lib.ptr(g.rowptr), _lib.ptr(g.colidx), _lib.ptr(g.t_rowptr),
                                     _lib.ptr(g.t_colidx), _lib.ptr(g.t_to_fwd), n, d, _lib.ptr(h),
                                     _lib.ptr(a_src), _lib.ptr(a_dst), ctx.slope, _lib.ptr(alpha),
                                     _lib.ptr(gout), _lib.ptr(gh), _lib.ptr(ga_src), _lib.ptr(ga_dst),
                                     _lib.ptr(dpre), _lib.stream()), "gda_gat_bwd_f32")
        return gh, ga_src, ga_dst, None, None


class GATConv(nn.Module):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0.0,
                 add_self_loops=True, bias=True, **kwargs):
        super().__init__()
        if heads != 1 or dropout != 0.0 or not add_self_loops:
            raise NotImplementedError("GNNBase uses GATConv(heads=1, concat=False) with PyG's other defaults")
        self.in_channels, self.out_channels, self.negative_slope = in_channels, out_channels, negative_slope
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        self.att_src = nn.Parameter(torch.empty(1, 1, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, 1, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        glorot(self.att_src)
        glorot(self.att_dst)
        zeros(self.bias)

    def forward(self, x, edge_index, edge_attr=None, size=None):
        h = self.lin(x)
        a_src = (h * self.att_src.view(1, -1)).sum(-1)
        a_dst = (h * self.att_dst.view(1, -1)).sum(-1)
        out = _GatAggregate.apply(h, a_src, a_dst, _gat_graph(edge_index, x.size(0)), self.negative_slope)
        return out + self.bias if self.bias is not None else out
