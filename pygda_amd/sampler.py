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

    def sample(self, seeds, fanouts, seed=0, csr=False):
        """-> (n_id [n_nodes] global ids, seeds first; edge_index [2, n_edges] local ids); with ``csr`` a third
        item: the GCN-normalised adjacency of the batch as one int32 host block ``[rowptr | t_rowptr | colidx |
        t_colidx | val bits | t_val bits]`` (gda_sampler_csr_norm: what the device ingestion would build from
        ``edge_index``, without its sorts)."""
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
        if not csr:
            return torch.from_numpy(nodes), torch.from_numpy(ei)
        n = nn_.value
        o = self._csr_offsets(n, ne_.value)
        block = np.empty(o[6], dtype=np.int32)
        at = lambda k: block[o[k]:].ctypes.data
        _lib.check(L.gda_sampler_csr_norm(self._h, at(0), at(2), at(4), at(1), at(3), at(5)), "gda_sampler_csr_norm")
        return torch.from_numpy(nodes), torch.from_numpy(ei), torch.from_numpy(block)

    def sample_batch(self, data, seeds, fanouts, seed=0):
        """A ``Data`` batch like PyG's: ``x``/``y`` sliced to the sampled nodes (seeds are the
        first ``batch_size`` rows), local ``edge_index``, ``n_id``, ``batch_size``."""
        if data.x.device.type == "cuda":
            return self.assemble_packed(data, seeds, self.sample_packed(seeds, fanouts, seed, csr=self.emit_csr(data)))
        return self.assemble(data, seeds, *self.sample(seeds, fanouts, seed))

    @staticmethod
    def emit_csr(data):
        """The host-built CSR rides along for batches assembled on the GPU (``PYGDA_AMD_SAMPLER_CSR=0``: off)."""
        import os
        return data.x.device.type == "cuda" and os.environ.get("PYGDA_AMD_SAMPLER_CSR", "1") == "1"

    @staticmethod
    def _csr_offsets(n, e):
        """Word offsets of [rowptr | t_rowptr | colidx | t_colidx | val | t_val] in a batch's CSR block, every
        array on a 16-byte boundary; the seventh entry is the block's size."""
        r4 = lambda v: (v + 3) // 4 * 4
        rp, cap = r4(n + 1), r4(n + e)
        return [0, rp, 2 * rp, 2 * rp + cap, 2 * rp + 2 * cap, 2 * rp + 3 * cap, 2 * rp + 4 * cap]

    # -- one pinned block per batch: ids, edges and the CSR cross PCIe in ONE asynchronous copy --------------
    def _pinned(self, nbytes):
        """A free pinned byte block of at least ``nbytes`` from this sampler's ring (blocks are recycled once the
        copy that read them has completed; a pageable source would make every ``.to(device)`` a blocking staged
        copy in the training thread: measured 3 ms per cfg-S step)."""
        import threading
        if not hasattr(self, "_ring"):
            self._ring, self._ring_lock = [], threading.Lock()
        with self._ring_lock:
            for ent in self._ring:
                if not ent["busy"] and ent["buf"].numel() >= nbytes and (ent["event"] is None or ent["event"].query()):
                    ent["busy"] = True
                    return ent
            ent = dict(buf=torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory(), event=None, busy=True)
            self._ring.append(ent)
            if len(self._ring) > 16:              # sizes drifted upwards: drop the smallest idle block
                idle = [x for x in self._ring if not x["busy"] and (x["event"] is None or x["event"].query())]
                if idle:
                    self._ring.remove(min(idle, key=lambda x: x["buf"].numel()))
            return ent

    def sample_packed(self, seeds, fanouts, seed=0, csr=True, device=None):
        """Sample a batch straight into one pinned block: ``[n_id int64 | esrc int64 | edst int64 | CSR block]``
        (the CSR part as in :meth:`sample`).  Runs on the loader's producer thread; with ``device`` the block is
        shipped from there as well, on this sampler's copy stream -- the copy (9 MB per cfg-S batch) then runs
        beside the previous step's kernels instead of in front of this step's on the training stream."""
        seeds = np.ascontiguousarray(torch.as_tensor(seeds).cpu().numpy(), dtype=np.int64)
        fan = np.ascontiguousarray(np.asarray(fanouts, dtype=np.int32))
        nn_, ne_ = ctypes.c_int64(), ctypes.c_int64()
        L = _lib.lib()
        _lib.check(L.gda_sampler_sample(self._h, seeds.ctypes.data, seeds.size, fan.ctypes.data, fan.size,
                                        ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), ctypes.byref(nn_),
                                        ctypes.byref(ne_)), "gda_sampler_sample")
        n, e = nn_.value, ne_.value
        r16 = lambda v: (v + 15) // 16 * 16
        o_nodes, o_edges = 0, r16(8 * n)
        o_csr = o_edges + r16(16 * e)
        co = self._csr_offsets(n, e)
        total = o_csr + (4 * co[6] if csr else 0)
        ent = self._pinned(max(total, 16))
        base = ent["buf"].data_ptr()
        _lib.check(L.gda_sampler_fetch(self._h, base + o_nodes, base + o_edges if e else None,
                                       base + o_edges + 8 * e if e else None), "gda_sampler_fetch")
        if csr:
            at = lambda k: base + o_csr + 4 * co[k]
            _lib.check(L.gda_sampler_csr_norm(self._h, at(0), at(2), at(4), at(1), at(3), at(5)), "gda_sampler_csr_norm")
            nnz = int(ent["buf"][o_csr + 4 * n:o_csr + 4 * n + 4].view(torch.int32)[0])     # rowptr[n]
        else:
            nnz = None
        pk = dict(ent=ent, n=n, e=e, total=total, o_nodes=o_nodes, o_edges=o_edges, o_csr=o_csr, csr=csr, nnz=nnz)
        if device is not None:
            self._ship(pk, torch.device(device))
        return pk

    def _ship(self, pk, dev):
        """The one H2D copy of a packed batch, on the copy stream of the calling (producer) thread."""
        if getattr(self, "_copy_stream", None) is None:
            torch.cuda.set_device(dev)
            self._copy_stream = torch.cuda.Stream(device=dev)
        ent = pk["ent"]
        with torch.cuda.stream(self._copy_stream):
            pk["dev"] = ent["buf"][:max(pk["total"], 16)].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        pk["ready"] = ev
        with self._ring_lock:
            ent["event"], ent["busy"] = ev, False

    def assemble_packed(self, data, seeds, pk):
        """Device side of a packed batch: ONE asynchronous copy of the pinned block, typed views of its device
        image, feature rows by the gather kernel."""
        from .graph import CSRGraph
        from .ops import gather_rows
        dev = data.x.device
        ent, n, e = pk["ent"], pk["n"], pk["e"]
        if "dev" in pk:                       # shipped by the producer on its copy stream
            b = pk["dev"]
            torch.cuda.current_stream().wait_event(pk["ready"])
            b.record_stream(torch.cuda.current_stream())
        else:
            b = ent["buf"][:max(pk["total"], 16)].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            with self._ring_lock:
                ent["event"], ent["busy"] = ev, False
        n_dev = b[pk["o_nodes"]:pk["o_nodes"] + 8 * n].view(torch.int64)
        ei = b[pk["o_edges"]:pk["o_edges"] + 16 * e].view(torch.int64).view(2, e)
        ei._gda_trusted = True                # relabelled ids are in range by construction: no validation sync
        if pk["csr"]:                         # gcn_norm(edge_index) of this batch, ready made (graph.as_graph)
            co, cap = self._csr_offsets(n, e), n + e
            sizes = [n + 1, n + 1, cap, cap, cap, cap]
            w = b[pk["o_csr"]:].view(torch.int32)
            part = lambda k: w[co[k]:co[k] + max(sizes[k], 1)]
            g = CSRGraph(n, cap, part(0), part(2), part(4).view(torch.float32), part(1), part(3), part(5).view(torch.float32))
            g._nnz, g.transient = pk["nnz"], True
            ei._gda_prebuilt = g
        x = gather_rows(data.x, n_dev)
        y = None if data.y is None else data.y[n_dev]
        return Data(x=x, edge_index=ei, y=y, n_id=n_dev, batch_size=int(torch.as_tensor(seeds).numel()))

    def assemble(self, data, seeds, n_id, ei):
        """A batch from host tensors (CPU data; or device data without the packed path)."""
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
