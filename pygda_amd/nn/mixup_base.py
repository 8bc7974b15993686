"""``MixupBase`` (pygda/nn/mixup_base.py:10-200): StruRW's mixup backbone.  Per layer the reference runs three
MixUpGCNConv calls -- ``(x, x)`` on the graph, ``(x, x_mix)`` on the graph, ``(x[perm], x_mix)`` on the shuffled
graph -- and mixes the last two: ``x_mix' = drop(lam act(new) + (1 - lam) act(new_b))``.

MI355X mapping (csrc/gda_mixup.hip): the shuffled graph StruRW builds (strurw.py:702-758) is the SAME graph with
its nodes renumbered, so its aggregate is a row permutation of the first one: one aggregation per layer instead
of three, one centre GEMM over the stacked pair ``[x ; x_mix]``, and one fused epilogue launch each way for
everything else.  A foreign ``edge_index_b`` (any tensor that is not known to be that renumbering) gets its own
aggregation and the same epilogue."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..ops import gather_rows, mixup_combine, mixup_combine_ok
from .mixup_gcnconv import MixupGraphCache, MixUpGCNConv
from .linear import DenseLinear


class ShuffledEdges:
    """What StruRW.shuffle_data hands over instead of a materialised ``edge_index_b``: the edge list it was
    derived from and the permutation (``id_new_value_old``).  ``tensor()`` builds the renumbered edge list
    (strurw.py:748-756) for consumers that want one."""

    def __init__(self, edge_index, id_new_value_old):
        self.edge_index, self.id_new_value_old = edge_index, id_new_value_old

    def tensor(self):
        perm = torch.as_tensor(self.id_new_value_old, dtype=torch.long, device=self.edge_index.device)
        old_to_new = torch.empty_like(perm)
        old_to_new[perm] = torch.arange(perm.numel(), device=perm.device)
        return old_to_new[self.edge_index]


class MixupBase(nn.Module):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=1, dropout=0.1, act=F.relu, rw_lmda=0.8, **kwargs):
        super().__init__()
        self.in_dim, self.hid_dim, self.num_classes, self.num_layers = in_dim, hid_dim, num_classes, num_layers
        self.dropout, self.act, self.rw_lmda = dropout, act, rw_lmda
        self.convs = nn.ModuleList([MixUpGCNConv(in_dim, hid_dim)] +
                                   [MixUpGCNConv(hid_dim, hid_dim) for _ in range(num_layers - 1)])
        self.cls = DenseLinear(hid_dim, num_classes)
        self._graphs = MixupGraphCache()                   # one CSR for all layers (same normalisation)

    def forward(self, x, edge_index, edge_index_b, lam, id_new_value_old, edge_weight):
        return self.feat_classifier(self.feat_bottleneck(x, edge_index, edge_index_b, lam, id_new_value_old, edge_weight))

    def feat_classifier(self, x):
        return self.cls(x)

    # ------------------------------------------------------------------ the layer --
    def _perm(self, id_new_value_old, n, device):
        """(perm, inverse, is_identity) as int64 device vectors; one upload when the permutation lives on the host."""
        if torch.is_tensor(id_new_value_old):
            perm = id_new_value_old.to(device=device, dtype=torch.long)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(n, device=device)
            return perm, inv, False
        host = np.asarray(id_new_value_old, dtype=np.int64)
        if host.shape != (n,):
            raise ValueError(f"id_new_value_old must list {n} nodes, got shape {host.shape}")
        cached = getattr(self, "_identity", None)
        if np.array_equal(host, np.arange(n)):
            if cached is None or cached.numel() != n or cached.device != device:
                cached = self._identity = torch.arange(n, device=device)
            return cached, cached, True
        inv = np.empty_like(host)
        inv[host] = np.arange(n)
        both = torch.from_numpy(np.stack([host, inv])).to(device)
        return both[0], both[1], False

    def _layer(self, conv, xs, centre_in, first, graph, graph_b, perm, inv, lam):
        """One layer on ``xs`` (the plain rows) -> the stacked pair ``[x' ; x_mix']``."""
        n = xs.size(0)
        P = conv.aggregate(xs, graph)
        Pb = None if graph_b is None else conv.aggregate(gather_rows(xs, perm) if xs.is_cuda else xs[perm], graph_b)
        CC = conv.lin_cen(centre_in)                       # [n, h] (first) or [2n, h]
        if self.act is F.relu and mixup_combine_ok(P, P.size(1)) and conv.bias is not None:
            return mixup_combine(P, Pb, CC, conv.bias, perm, inv, lam, self.dropout, self.training, first)
        C, Cm = (CC, lam * CC + (1 - lam) * CC[perm]) if first else (CC[:n], CC[n:])
        b = 0 if conv.bias is None else conv.bias
        Pb = P[perm] if Pb is None else Pb
        xn = F.dropout(self.act(P + C + b), p=self.dropout, training=self.training)
        mix = self.act(P + Cm + b) * lam + self.act(Pb + Cm + b) * (1 - lam)
        return torch.cat([xn, F.dropout(mix, p=self.dropout, training=self.training)])

    def feat_bottleneck(self, x, edge_index, edge_index_b, lam, id_new_value_old, edge_weight):
        if len(self.convs) < 2:
            raise IndexError("MixupBase needs num_layers >= 2 (mixup_base.py:150 reads convs[1])")
        n = x.size(0)
        perm, inv, identity = self._perm(id_new_value_old, n, x.device)
        graph = self._graphs.get(edge_index, edge_weight, self.rw_lmda, n)
        renumbering = (isinstance(edge_index_b, ShuffledEdges) and edge_index_b.edge_index is edge_index
                       and edge_index_b.id_new_value_old is id_new_value_old) or (edge_index_b is edge_index and identity)
        if renumbering:
            graph_b = None                                 # Aggb(lin(x[perm])) = Agg(lin(x))[perm]
        else:
            if isinstance(edge_index_b, ShuffledEdges):
                edge_index_b = edge_index_b.tensor()
            graph_b = self._graphs.get(edge_index_b, edge_weight, self.rw_lmda, n)
        lam = float(lam)
        XX = self._layer(self.convs[0], x, x, True, graph, graph_b, perm, inv, lam)
        for conv in self.convs[1:]:
            XX = self._layer(conv, XX[:n], XX, False, graph, graph_b, perm, inv, lam)
        return XX[n:]
