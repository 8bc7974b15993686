"""``PPMIConv`` (pygda/nn/ppmi_conv.py:10-184): a ``CachedGCNConv`` whose cached graph is
the random-walk PPMI graph of the input graph.  The minutes-long Python walk loop of the
reference becomes one call into the device builder (csrc/gda_ppmi_dev.hip: walks, sorts and
run-length counts on the GPU) or, for edge lists on the host and graphs beyond its int32 limits,
the native host builder (csrc/gda_ppmi.cpp) -- same estimator, same counter-based walks; self
loops and the source-degree normalisation run in the device ingestion kernel."""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..graph import CSRGraph, build_csr
from .cached_gcn_conv import CachedGCNConv


def ppmi_edges_device(edge_index, num_nodes, path_len, passes, seed):
    """Device builder; None when the graph exceeds its int32 limits (caller falls back to the host)."""
    L = _lib.lib()
    E, N = int(edge_index.size(1)), int(num_nodes)
    need = L.gda_ppmi_workspace_bytes(E, N, int(path_len), int(passes))
    if need == 0:
        return None
    dev = edge_index.device
    cap = max(N * int(passes) * int(path_len), 1)
    out_ei = torch.empty(2, cap, dtype=torch.int64, device=dev)
    out_w = torch.empty(cap, dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)          # one-off, hundreds of MB: not cached
    src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
    _lib.check(L.gda_ppmi_build(_lib.ptr(src), _lib.ptr(dst), E, N, int(path_len), int(passes),
                                ctypes.c_uint64(seed & (2 ** 64 - 1)), _lib.ptr(out_ei[0]), _lib.ptr(out_ei[1]),
                                _lib.ptr(out_w), _lib.ptr(count), _lib.ptr(ws), ws.numel(), _lib.stream()),
               "gda_ppmi_build")
    m = int(count.item())                                           # the one data-dependent size, once per graph
    return out_ei[:, :m].clone(), out_w[:m].clone()


def ppmi_edges(edge_index, num_nodes, path_len=5, passes=40, seed=None, device_builder=True):
    """Weighted PPMI edge list ``(edge_index [2, M] int64, weight [M] fp32)``, on the device of
    ``edge_index`` when that is a GPU (device builder), else on the CPU (host builder).
    ``seed=None`` draws the seed from ``np.random`` so that ``np.random.seed`` governs the
    result as it does in the reference (the stream itself differs: see gda_ppmi.cpp)."""
    if seed is None:
        seed = (int(np.random.randint(0, 2 ** 31)) << 31) | int(np.random.randint(0, 2 ** 31))
    if device_builder and edge_index.is_cuda:
        got = ppmi_edges_device(edge_index, num_nodes, path_len, passes, seed)
        if got is not None:
            return got
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
