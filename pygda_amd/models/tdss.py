"""``TDSS`` trainer (pygda/models/tdss.py:93-661): A2GNN's asymmetric-propagation encoder and
sampled MMD, plus a degree-normalised Laplacian smoothness term of the target features over a
K-hop / random-walk "smoothing" graph built once in ``fit``.

MI355X mapping: the smoothing graph is built by the native host builders (csrc/gda_smooth.cpp;
the reference goes through ``spspmm`` or a dense N x N matrix), ingested once into CSR, and the
loss + its gradient are two gather kernels (csrc/gda_laplacian.hip) instead of an autograd chain
over four ``[E_smooth, d]`` temporaries."""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib
from ..ops import laplacian_loss
from ..utils import MMD
from .a2gnn import A2GNN


def _fetch_edge_list(handle):
    L = _lib.lib()
    try:
        m = L.gda_edge_list_size(handle)
        out = np.empty((2, m), dtype=np.int64)
        _lib.check(L.gda_edge_list_fetch(handle, out[0].ctypes.data if m else None,
                                         out[1].ctypes.data if m else None, None), "gda_edge_list_fetch")
    finally:
        L.gda_edge_list_destroy(handle)
    return torch.from_numpy(out)


def _host_edges(edge_index):
    ei = edge_index.detach().cpu().numpy()
    return np.ascontiguousarray(ei[0], dtype=np.int64), np.ascontiguousarray(ei[1], dtype=np.int64)


def _threads():
    return max(1, min(32, os.cpu_count() or 1))


def two_hop_edges(edge_index, num_nodes, rounds=1):
    """``rounds`` applications of TwoHopNeighbor (tdss.py:67-87) to a bare edge list:
    ``coalesce(E U pattern(A.A) minus self loops)``, sorted by (row, col)."""
    src, dst = _host_edges(edge_index)
    h = ctypes.c_void_p()
    _lib.check(_lib.lib().gda_two_hop_host(src.ctypes.data, dst.ctypes.data, src.size, int(num_nodes),
                                           int(rounds), _threads(), ctypes.byref(h)), "gda_two_hop_host")
    return _fetch_edge_list(h).to(edge_index.device)


def walk_smooth_edges(edge_index, num_nodes, walk_len, seed=None):
    """Random-walk smoothing graph (tdss.py:367-373): edge (visited, start) for one uniform walk
    per node.  The seed is drawn from torch's CPU generator, so ``torch.manual_seed`` fixes it."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    src, dst = _host_edges(edge_index)
    h = ctypes.c_void_p()
    _lib.check(_lib.lib().gda_walk_smooth_host(src.ctypes.data, dst.ctypes.data, src.size, int(num_nodes),
                                               int(walk_len), seed, _threads(), ctypes.byref(h)),
               "gda_walk_smooth_host")
    return _fetch_edge_list(h).to(edge_index.device)


def _add_remaining_self_loops(edge_index, num_nodes):
    """PyG ``add_remaining_self_loops`` without attributes: existing loops dropped, one loop per
    node appended last."""
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index[:, keep], torch.stack([loops, loops])], dim=1)


class TwoHopNeighbor:
    """The graph transform of tdss.py:22-90 for objects with ``edge_index`` / ``num_nodes``
    (edge attributes, which TDSS never has, are not carried)."""

    def __call__(self, data):
        if getattr(data, "edge_attr", None) is not None:
            raise NotImplementedError("TwoHopNeighbor with edge attributes is outside the TDSS path")
        data.edge_index = two_hop_edges(data.edge_index, data.num_nodes, 1)
        return data

    def __repr__(self):
        return '{}()'.format(self.__class__.__name__)


class TDSS(A2GNN):
    def __init__(self, in_dim, hid_dim, num_classes, mode='node', smooth_mode='RW', num_layers=2,
                 dropout=0., act=F.relu, s_pnums=0, t_pnums=30, k=2, rw_len=4, alpha=0.001, beta=1e-4,
                 weight_decay=0.005, adv=False, lr=0.01, epoch=200, device='cuda:0', batch_size=0,
                 num_neigh=-1, verbose=2, **kwargs):
        assert mode == 'node', 'TDSS only supports node-level tasks'                     # tdss.py:192-193
        assert adv == False, 'TDSS does not support adversarial training'
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, mode=mode,
                         num_layers=num_layers, dropout=dropout, act=act, s_pnums=s_pnums, t_pnums=t_pnums,
                         adv=adv, weight=alpha, weight_decay=weight_decay, lr=lr, epoch=epoch, device=device,
                         batch_size=batch_size, num_neigh=num_neigh, verbose=verbose, **kwargs)
        self.smooth_mode, self.k, self.rw_len, self.alpha, self.beta = smooth_mode, k, rw_len, alpha, beta

    def smoothness(self, edge_index, edge_attr, num_nodes):
        """tdss.py:314-388 -> (edge_index_smooth, edge_attr_smooth)."""
        if self.smooth_mode == 'RW':
            ei = walk_smooth_edges(edge_index, num_nodes, self.rw_len)
            return ei, torch.ones(ei.size(1), device=ei.device)          # dense_to_sparse values
        if edge_attr is not None:
            raise NotImplementedError("K-hop smoothing with edge attributes is outside the TDSS path")
        if self.k == 1:                  # :375-376 passes no num_nodes: PyG infers max index + 1
            inferred = int(edge_index.max()) + 1 if edge_index.numel() else 0
            return _add_remaining_self_loops(edge_index, inferred), None
        return _add_remaining_self_loops(two_hop_edges(edge_index, num_nodes, self.k - 1), num_nodes), None

    def compute_laplacian_loss(self, features, edge_index):
        return laplacian_loss(features, edge_index)

    def forward_model(self, source_data, target_data, alpha):
        """tdss.py:241-312: source CE + alpha * MMD + beta * Laplacian(target features)."""
        net = self.a2gnn
        smooth = getattr(target_data, "edge_index_smooth", None)
        if smooth is None:
            raise ValueError("target_data.edge_index_smooth is missing: TDSS.fit() attaches it")
        if getattr(target_data, "n_id", None) is not None:
            raise NotImplementedError(
                "TDSS with sampled mini-batches: the reference indexes the batch's features with the "
                "smoothing graph of the WHOLE target graph (tdss.py:306) -- only full-batch training "
                "(batch_size=0) is well defined")
        loss, source_logits, source_features, target_features, h0_t, pending, _ = \
            self._branches(source_data, target_data)                                      # :275-287, three streams
        loss = loss + self.alpha * MMD(source_features, target_features)                  # :301-303
        loss = loss + self.beta * self.compute_laplacian_loss(target_features, smooth)    # :306-307
        if pending is not None:
            target_logits, side = pending
            torch.cuda.current_stream().wait_stream(side)
        elif self.compute_target_logits:                                                  # :309
            feats_t = net.feat_bottleneck_from(h0_t, target_data.edge_index, None, self.t_pnums)
            target_logits = net.feat_classifier(feats_t, target_data.edge_index, None, 1)
        else:
            target_logits = None
        return loss, source_logits, target_logits

    @property
    def _dp_graph_parts(self):
        raise AttributeError("TDSS has no segmented data-parallel step (hasattr() is the probe)")

    def fit(self, source_data, target_data):
        n_t = target_data.x.shape[0]
        target_data.edge_index_smooth, target_data.edge_attr_smooth = self.smoothness(
            target_data.edge_index, getattr(target_data, "edge_attr", None), n_t)          # :497
        if hasattr(target_data, "_device_copies"):
            target_data._device_copies.clear()           # device copies made earlier lack the new attributes
        self._train_epochs(*self._prepare(source_data, target_data))
