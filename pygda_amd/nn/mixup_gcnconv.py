"""``MixUpGCNConv`` (pygda/nn/mixup_gcnconv.py:69-247): ``out = Agg(lin(x)) + lin_cen(x_cen) + bias`` where the
aggregation runs over the unit-weight gcn normalisation WITHOUT self loops (:204-214; the ``edge_weight`` it is
handed is the structure re-weighting factor ``rw_e``, :200-201) and each message is
``((1 - lmda) + lmda rw_e) * norm_e * lin(x)_j`` (:242-245), summed at ``edge_index[1]``.

MI355X mapping: coefficient and normalisation fold into the values of one CSR, rebuilt only when the weights
change; one aggregation launch per call."""
import torch
from torch import nn

from ..graph import build_csr
from ..ops import propagate
from .linear import Linear, zeros


class MixupGraphCache:
    """CSR of ``((1-lmda) + lmda rw_e) d^-1/2[src] d^-1/2[dst]`` keyed on the identity / version of the edge
    tensors (kept alive by the entry).  A few entries: StruRW alternates between the re-weighted source graph,
    the unit-weight source graph its predict() puts back every epoch, and the target graph."""
    ENTRIES = 4

    def __init__(self):
        self._entries = []                                 # [(key, graph, hold)], most recent last

    def get(self, edge_index, edge_rw, lmda, n):
        if edge_rw is None:                                # the reference fails on None (:243); unit weights here
            key = (edge_index.data_ptr(), edge_index._version, None, None, float(lmda), n)
        else:
            key = (edge_index.data_ptr(), edge_index._version, edge_rw.data_ptr(), edge_rw._version, float(lmda), n)
        for k, (ekey, graph, _) in enumerate(self._entries):
            if ekey == key:
                self._entries.append(self._entries.pop(k))
                return graph
        src, dst = edge_index[0], edge_index[1]
        deg = torch.bincount(dst, minlength=n).to(torch.float32)
        dis = deg.pow(-0.5)
        dis.masked_fill_(dis == float("inf"), 0)
        val = dis[src] * dis[dst]
        if edge_rw is not None:
            val = val * ((1.0 - lmda) + lmda * edge_rw.detach().to(torch.float32))
        graph = build_csr(edge_index, n, val, add_self_loops=False, normalize=False)
        self._entries.append((key, graph, (edge_index, edge_rw)))
        del self._entries[:-self.ENTRIES]
        return graph


class MixUpGCNConv(nn.Module):
    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=False,
                 normalize=True, bias=True, **kwargs):
        super().__init__()
        if improved or add_self_loops or not normalize or kwargs.get("aggr", "add") != "add":
            raise NotImplementedError("MixUpGCNConv: only the configuration MixupBase builds is covered")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        self.lin_cen = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._graphs = MixupGraphCache()

    def reset_parameters(self):                            # :135-139: lin only, lin_cen keeps its construction draw
        self.lin.reset_parameters()
        zeros(self.bias)

    def aggregate(self, x, graph):
        """``Agg(lin(x))`` on a prepared graph (the part MixupBase shares between its three calls)."""
        return propagate(self.lin(x), graph, 1)

    def forward(self, x, x_cen, edge_index, edge_weight=None, lmda=1):
        graph = self._graphs.get(edge_index, edge_weight, lmda, x.size(0))
        out = self.aggregate(x, graph) + self.lin_cen(x_cen)
        return out if self.bias is None else out + self.bias

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"
