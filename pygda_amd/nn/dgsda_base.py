"""``BernProp`` / ``DGSDABase`` (pygda/nn/dgsda_base.py:11-315): Bernstein-polynomial spectral
filter with learnable coefficients, and the lin1 -> filter -> lin2 -> filter network of DGSDA.

The filter runs on the aggregation kernel with an affine epilogue (:func:`pygda_amd.ops.bern_filter`):
the symmetric normalisation without self loops is ingested once per graph, ``L x = x - A x`` and
``(2I - L) x = x + A x`` are single launches, and the K + K(K+1)/2 propagations of the reference
become 2K."""
from math import comb

import torch
import torch.nn.functional as F
from torch import nn

from .. import sparse_features
from ..graph import as_graph
from ..ops import bern_filter
from .linear import DenseLinear


class _InputLinear(nn.Linear):
    """``nn.Linear`` (same parameters, same init stream) whose product with a registered sparse
    input matrix -- the raw bag-of-words features -- runs on the aggregation kernel."""

    def forward(self, x, dropout=0.0):
        """``linear(dropout(x))``: for a registered sparse input the mask is applied to the stored
        non-zeros and the product stays sparse (no dense [N, F] mask pass, no dense GEMM)."""
        if x.dim() == 2 and x.size(1) >= sparse_features.MIN_WIDTH:
            sf = sparse_features.lookup(x)
            if sf is not None:
                return sparse_features.sparse_linear(self.weight, sf, dropout) + self.bias
        if dropout > 0.0:
            x = F.dropout(x, p=dropout, training=True)
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
            from .linear import tall_linear_bias             # the hand-written matrix-core kernels (no BLAS on the GPU path)
            return tall_linear_bias(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


class BernProp(nn.Module):
    def __init__(self, K, is_source_domain=True, bias=True, **kwargs):
        super().__init__()
        self.K = K
        self.is_source_domain = is_source_domain
        self.temp = nn.Parameter(torch.Tensor(self.K + 1), requires_grad=is_source_domain)
        self.register_buffer("_coefs", torch.tensor([comb(K, k) / (2 ** K) for k in range(K + 1)],
                                                    dtype=torch.float32), persistent=False)
        self.reset_parameters()

    def reset_parameters(self):                                              # dgsda_base.py:63-77
        if self.is_source_domain:
            self.temp.data.fill_(1)
        else:
            self.temp.data = torch.linspace(1, 0, self.K + 1)

    def forward(self, x, edge_index, edge_weight=None):
        # get_laplacian(..., 'sym') (:128): self loops removed, degree over `row`, A = D^-1/2 W D^-1/2
        graph = as_graph(edge_index, x.size(0), edge_weight, False, "drop", True, "row")
        return bern_filter(x, self.temp, graph, self._coefs)

    def __repr__(self):
        return '{}(K={}, temp={})'.format(self.__class__.__name__, self.K, self.temp)


class DGSDABase(nn.Module):
    def __init__(self, features, hidden, classes, dprate=0.0, K=15):
        super().__init__()
        self.lin1 = _InputLinear(features, hidden)
        self.lin2 = DenseLinear(hidden, classes)
        self.prop1 = BernProp(K)
        self.prop2 = BernProp(K)
        self.prop3 = BernProp(K)
        self.dprate = dprate

    def reset_parameters(self):
        self.prop1.reset_parameters()

    def forward(self, data, is_source_domain=True):                          # :238-276
        x, edge_index = data.x, data.edge_index
        x = self.get_props(x, edge_index, is_source_domain)
        x = F.dropout(x, p=self.dprate, training=self.training)
        x = self.lin2(x)
        x = F.dropout(x, p=self.dprate, training=self.training)
        return self.prop3(x, edge_index)

    def get_props(self, x, edge_index, is_source_domain=True):               # :278-315
        x = F.relu(self.lin1(x, self.dprate if self.training else 0.0))        # dropout folded into lin1
        x = F.dropout(x, p=self.dprate, training=self.training)
        return self.prop1(x, edge_index) if is_source_domain else self.prop2(x, edge_index)
