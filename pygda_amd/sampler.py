"""Python face of the native host neighbour sampler (csrc/gda_sampler.cpp) and the
mini-batch assembly around it: sample on the host, gather the feature rows on the device
(``gda_gather_rows_f32``) when the features live there."""
import ctypes

import numpy as np
import torch

from . import _lib
from .data import Data


class NeighborSampler:
    def __init__(self, edge_index, num_nodes, threads=None):
        ei = edge_index.detach().cpu().contiguous()
        self.num_nodes = int(num_nodes)
        self._src = np.ascontiguousarray(ei[0].numpy(), dtype=np.int64)
        self._dst = np.ascontiguousarray(ei[1].numpy(), dtype=np.int64)
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        _lib.check(L.gda_sampler_create(self._src.ctypes.data, self._dst.ctypes.data, self._src.size,
                                        self.num_nodes, ctypes.byref(self._h)), "gda_sampler_create")
        if threads is None:
            import os
            threads = max(1, min(16, (os.cpu_count() or 2) // 2))
        _lib.check(L.gda_sampler_set_threads(self._h, int(threads)), "gda_sampler_set_threads")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:   # not during interpreter teardown
            _lib._lib.gda_sampler_destroy(h)
            self._h = None

    def sample(self, seeds, fanouts, seed=0):
        """-> (n_id [n_nodes] global ids, seeds first; edge_index [2, n_edges] local ids)."""
        seeds = np.ascontiguousarray(torch.as_tensor(seeds).cpu().numpy(), dtype=np.int64)
        fan = np.ascontiguousarray(np.asarray(fanouts, dtype=np.int32))
        nn_, ne_ = ctypes.c_int64(), ctypes.c_int64()
        L = _lib.lib()
        _lib.check(L.gda_sampler_sample(self._h, seeds.ctypes.data, seeds.size, fan.ctypes.data, fan.size,
                                        ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), ctypes.byref(nn_),
                                        ctypes.byref(ne_)), "gda_sampler_sample")
        nodes = np.empty(nn_.value, dtype=np.int64)
        ei = np.empty((2, ne_.value), dtype=np.int64)
        _lib.check(L.gda_sampler_fetch(self._h, nodes.ctypes.data, ei[0].ctypes.data if ne_.value else None,
                                       ei[1].ctypes.data if ne_.value else None), "gda_sampler_fetch")
        return torch.from_numpy(nodes), torch.from_numpy(ei)

    def sample_batch(self, data, seeds, fanouts, seed=0):
        """A ``Data`` batch like PyG's: ``x``/``y`` sliced to the sampled nodes (seeds are the
        first ``batch_size`` rows), local ``edge_index``, ``n_id``, ``batch_size``."""
        n_id, ei = self.sample(seeds, fanouts, seed)
        return self.assemble(data, seeds, n_id, ei)

    def assemble(self, data, seeds, n_id, ei):
        """Device side of a batch: feature rows by the gather kernel, labels, ids."""
        dev = data.x.device
        if dev.type == "cuda":
            from .ops import gather_rows
            n_dev = n_id.to(dev, non_blocking=True)
            x = gather_rows(data.x, n_dev)
            y = None if data.y is None else data.y[n_dev]
            ei = ei.to(dev, non_blocking=True)
            ei._gda_trusted = True            # relabelled ids are in range by construction: no validation sync
            return Data(x=x, edge_index=ei, y=y, n_id=n_dev,
                        batch_size=int(torch.as_tensor(seeds).numel()))
        return Data(x=data.x[n_id], edge_index=ei, y=None if data.y is None else data.y[n_id], n_id=n_id,
                    batch_size=int(torch.as_tensor(seeds).numel()))
