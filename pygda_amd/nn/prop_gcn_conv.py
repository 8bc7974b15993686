"""``PropGCNConv`` / ``gcn_norm`` on the MI355X aggregation kernels.

Operator surface of pygda/nn/prop_gcn_conv.py:24-264 (same constructor, ``forward(x,
edge_index, prop_nums=1, edge_weight=None)``, ``.lin.weight [out, in]``, ``.bias``); the
body is ours: one cached CSR ingestion per graph instead of a ``gcn_norm`` per call, one
K-step SpMM launch sequence instead of ``prop_nums`` PyG propagates, and a backward that
re-runs the same kernel on the transposed CSR instead of saving K intermediates.
"""
import torch
from torch import nn

from .. import sparse_features
from ..graph import CSRGraph, as_graph, build_csr
from ..ops import propagate
from .linear import Linear, zeros

import os
SPARSE_COLMAJOR = os.environ.get("PYGDA_AMD_SPARSE_COLMAJOR", "1") == "1"


def gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False, add_self_loops=True,
             dtype=None):
    """Symmetric normalisation with self loops (prop_gcn_conv.py:24-81, tensor branch).

    Returns ``(edge_index, edge_weight)`` like the reference, listed by destination node
    (stable in the original edge order inside a destination) rather than in input order --
    the multiset of weighted edges is identical.
    """
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if edge_index.numel() else 0
    g = build_csr(edge_index, num_nodes, edge_weight, improved, add_self_loops, True, "col")
    return g.to_coo()


class PropGCNConv(nn.Module):
    def __init__(self, in_channels, out_channels, improved=False, cached=False,
                 add_self_loops=True, normalize=True, bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached = improved, cached
        self.add_self_loops, self.normalize = add_self_loops, normalize
        self._cached_graph = None
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()          # second glorot draw, as in the reference (:144-147)

    def reset_parameters(self):
        self.lin.reset_parameters()
        zeros(self.bias)
        self._cached_graph = None

    def _graph(self, x, edge_index, edge_weight):
        if isinstance(edge_index, CSRGraph):
            return edge_index
        if self.cached and self._cached_graph is not None:
            return self._cached_graph
        g = as_graph(edge_index, x.size(0), edge_weight, self.improved,
                     self.add_self_loops and self.normalize, self.normalize, "col")
        if self.cached:
            self._cached_graph = g
        return g

    def forward(self, x, edge_index, prop_nums=1, edge_weight=None):
        return self._forward(x, edge_index, prop_nums, edge_weight, False)

    def forward_colmajor(self, x, edge_index, prop_nums=1, edge_weight=None):
        """forward() whose result may come back as an :class:`~pygda_amd.ops.ColMajor` (the K-step kernel's own
        layout) for the fused activation kernel that follows a conv in A2GNNBase; not in the reference."""
        return self._forward(x, edge_index, prop_nums, edge_weight, True)

    def forward_stacked(self, x, p, training, mode):
        """``prop_nums = 0`` layers of a sampled batch with the activation in the projection's epilogue
        (ops.tall_linear_act): ``mode`` 2 = two dropout draws of ``relu(x W^T + b)`` stacked ``[2n, out]`` (layer 0 of the
        trainer's two source passes), 3 = activation, handed out as the stacked rows' two halves (their last layer), 1 =
        plain.  None when the shape is not the fused kernel's (the caller composes).  ``x``: tensor or GatheredRows."""
        from ..ops import tall_fused_ok, tall_linear_act
        if self.bias is None or not tall_fused_ok(x, self.lin.weight):
            return None
        return tall_linear_act(x, self.lin.weight, self.bias, p, training, mode)

    def forward_act(self, x, edge_index, prop_nums, p, training, pair=False):
        """``dropout(relu(forward(x, edge_index, prop_nums)), p, training)`` when the aggregation's own epilogue can
        apply the activation -- a sampled batch on the one-launch interior K-step (ops.propagate_act) -- else None (the
        caller composes).  ``pair``: two independent dropout draws of the same pre-activation ``(a, b)``; ``b`` is not
        differentiated (the trainer's loss-unused second pass).  Not in the reference."""
        from ..ops import GatheredRows, propagate_act, propagate_act_ok, tall_fused_ok, tall_linear_act
        gathered = isinstance(x, GatheredRows)
        if prop_nums <= 0 or not (isinstance(x, torch.Tensor) or gathered) or not x.is_cuda or x.dim() != 2:
            return None
        g = self._graph(x, edge_index, None)
        if getattr(g, "n_interior", None) is None or getattr(g, "iplan", None) is None or self.out_channels % 4:
            return None
        if gathered:             # the batch's feature gather rides in the projection's operand fetch when it can
            out = tall_linear_act(x, self.lin.weight, None, 0.0, False, 0) if tall_fused_ok(x, self.lin.weight) \
                else self.lin(x.dense())
        else:
            out = self.lin(x)                                    # :205
        if not propagate_act_ok(out, g, prop_nums, self.bias):
            from ..ops import relu_dropout
            pre = propagate(out, g, prop_nums, self.bias)
            a = relu_dropout(pre, p, training)
            return (a, relu_dropout(pre.detach(), p, training)) if pair else a
        return propagate_act(out, g, prop_nums, self.bias, p, training, pair)

    def _forward(self, x, edge_index, prop_nums, edge_weight, colmajor_out):
        from ..ops import GatheredRows
        if isinstance(x, GatheredRows):      # (only the fused sampled-batch paths above read a batch's rows through its ids)
            x = x.dense()
        if colmajor_out and prop_nums > 0 and self.lin.tall_gemm_ok(x):
            # dense projection on the matrix-core kernels: it can write the K-step kernel's column-major
            # layout itself (and read the column-major gradient), so no transposition is left around the
            # aggregation of this layer
            from ..ops import lds_kstep_plan, tall_linear_colmajor
            g = self._graph(x, edge_index, edge_weight)
            if (self.out_channels % 4 == 0 and lds_kstep_plan(g, prop_nums, False) is not None
                    and lds_kstep_plan(g, prop_nums, True) is not None):
                return propagate(tall_linear_colmajor(x, self.lin.weight), g, prop_nums, self.bias)
        if (colmajor_out and prop_nums > 0 and x.dim() == 2 and x.size(1) >= sparse_features.MIN_WIDTH
                and self.out_channels % 4 == 0 and SPARSE_COLMAJOR):
            # sparse input features: the SpMM that projects them writes the K-step kernel's layout itself
            sf = sparse_features.lookup(x)
            if sf is not None:
                from ..ops import lds_kstep_plan
                g = self._graph(x, edge_index, edge_weight)
                if lds_kstep_plan(g, prop_nums, False) is not None and lds_kstep_plan(g, prop_nums, True) is not None:
                    hT = sparse_features.sparse_linear_colmajor(self.lin.weight, sf)
                    if hT is not None:
                        return propagate(hT, g, prop_nums, self.bias)
        if prop_nums <= 0 and self.bias is not None and self.lin.tall_gemm_ok(x):
            from .linear import tall_linear_bias               # projection + bias (and both gradients) in the GEMMs
            return tall_linear_bias(x, self.lin.weight, self.bias)
        if prop_nums <= 0 and self.bias is not None and x.dim() == 2 and x.size(1) >= sparse_features.MIN_WIDTH:
            sf = sparse_features.lookup(x)                       # sparse input features: bias in the SpMM's epilogue
            if sf is not None:
                return sparse_features.sparse_linear(self.lin.weight, sf, bias=self.bias)
        out = self.lin(x)                                        # :205
        if prop_nums > 0:                                        # :208-213, bias fused in the last step
            return propagate(out, self._graph(x, edge_index, edge_weight), prop_nums, self.bias,
                             colmajor_out=colmajor_out)
        if self.bias is not None:
            out = out + self.bias
        return out

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"
