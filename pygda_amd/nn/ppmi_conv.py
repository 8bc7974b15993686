"""``PPMIConv`` (pygda/nn/ppmi_conv.py:10-184): a ``CachedGCNConv`` whose cached graph is
the random-walk PPMI graph of the input graph.  The minutes-long Python walk loop of the
reference becomes one call into the native host builder (csrc/gda_ppmi.cpp); self loops and
the source-degree normalisation run in the device ingestion kernel."""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..graph import CSRGraph, build_csr
from .cached_gcn_conv import CachedGCNConv


def ppmi_edges(edge_index, num_nodes, path_len=5, passes=40, seed=None):
    """Weighted PPMI edge list ``(edge_index [2, M] int64, weight [M] fp32)`` on the CPU.
    ``seed=None`` draws the seed from ``np.random`` so that ``np.random.seed`` governs the
    result as it does in the reference (the stream itself differs: see gda_ppmi.cpp)."""
    if seed is None:
        seed = (int(np.random.randint(0, 2 ** 31)) << 31) | int(np.random.randint(0, 2 ** 31))
    ei = edge_index.detach().cpu().contiguous()
    src = np.ascontiguousarray(ei[0].numpy(), dtype=np.int64)
    dst = np.ascontiguousarray(ei[1].numpy(), dtype=np.int64)
    L = _lib.lib()
    h = ctypes.c_void_p()
    _lib.check(L.gda_ppmi_build_host(src.ctypes.data, dst.ctypes.data, src.size, int(num_nodes), int(path_len),
                                     int(passes), ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.byref(h)),
               "gda_ppmi_build_host")
    try:
        m = L.gda_edge_list_size(h)
        out_ei, out_w = np.empty((2, m), dtype=np.int64), np.empty(m, dtype=np.float32)
        _lib.check(L.gda_edge_list_fetch(h, out_ei[0].ctypes.data if m else None,
                                         out_ei[1].ctypes.data if m else None,
                                         out_w.ctypes.data if m else None), "gda_edge_list_fetch")
    finally:
        L.gda_edge_list_destroy(h)
    return torch.from_numpy(out_ei), torch.from_numpy(out_w)


class PPMIConv(CachedGCNConv):
    def __init__(self, in_channels, out_channels, weight=None, bias=None, improved=False, use_bias=True,
                 path_len=5, **kwargs):
        super().__init__(in_channels, out_channels, weight, bias, improved, use_bias, **kwargs)
        self.path_len = path_len

    def _ppmi_graph(self, edge_index, num_nodes, improved):
        ei, w = ppmi_edges(edge_index, num_nodes, self.path_len)
        dev = edge_index.device
        return build_csr(ei.to(dev), num_nodes, w.to(dev), improved, True, True, "row")

    def norm(self, edge_index, num_nodes, edge_weight=None, improved=False, dtype=None):
        """(edge_index, weight) of the normalised PPMI graph incl. self loops (:56-184)."""
        return self._ppmi_graph(edge_index, num_nodes, improved).to_coo()

    def _graph(self, x, edge_index, cache_name, edge_weight):
        if isinstance(edge_index, CSRGraph):
            return edge_index
        g = self.cache_dict.get(cache_name)
        if g is None:
            g = self._ppmi_graph(edge_index, x.size(0), self.improved)
            self.cache_dict[cache_name] = g
        return g
