"""Python face of the native host neighbour sampler (csrc/gda_sampler.cpp) and the
mini-batch assembly around it: sample on the host, gather the feature rows on the device
(``gda_gather_rows_f32``) when the features live there."""
import ctypes

import numpy as np
import torch

from . import _lib
from .data import Data


def default_threads():
    """Worker threads of ONE host sampler.  A rank runs two loaders (source and target) that prefetch at the same
    time, and a node runs LOCAL_WORLD_SIZE ranks: the host's hardware threads (this process's affinity mask) are
    divided by both, capped at 16 -- 8 ranks x 2 loaders x 8 threads on a 128-thread host, not 256 threads
    (``PYGDA_AMD_SAMPLER_THREADS`` overrides)."""
    import os
    env = os.environ.get("PYGDA_AMD_SAMPLER_THREADS")
    if env:
        return max(1, int(env))
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 2
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return max(1, min(16, cpus // (2 * local_world)))


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
        self.threads = default_threads() if threads is None else int(threads)
        _lib.check(L.gda_sampler_set_threads(self._h, self.threads), "gda_sampler_set_threads")

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

    def description(self):
        return f"native host sampler (csrc/gda_sampler.cpp), {self.threads} worker threads per loader"

    def sample_batch(self, data, seeds, fanouts, seed=0):
        """A ``Data`` batch like PyG's: ``x``/``y`` sliced to the sampled nodes (seeds are the
        first ``batch_size`` rows), local ``edge_index``, ``n_id``, ``batch_size``."""
        return self.assemble(data, seeds, *self.sample(seeds, fanouts, seed, csr=self.emit_csr(data)))

    @staticmethod
    def emit_csr(data):
        """``PYGDA_AMD_SAMPLER_CSR=1``: the batch's normalised CSR pair is built by the sampler (gda_sampler_csr_norm)
        and rides along instead of being re-derived by the device ingestion.  Off by default: measured at cfg-S
        (5 M nodes per domain, fan-out [15, 10], ~160 k-node batches) the device sorts it saves cost 0.5 ms per step,
        the 6 MB of extra host-to-device traffic per batch more than that (8.6 -> 9.4 ms per step; shipping ids,
        edges and CSR as one pinned block on a copy stream from the producer thread: 11.2 -> 13.4 ms)."""
        import os
        return data.x.device.type == "cuda" and os.environ.get("PYGDA_AMD_SAMPLER_CSR", "0") == "1"

    @staticmethod
    def _csr_offsets(n, e):
        """Word offsets of [rowptr | t_rowptr | colidx | t_colidx | val | t_val] in a batch's CSR block, every
        array on a 16-byte boundary; the seventh entry is the block's size."""
        r4 = lambda v: (v + 3) // 4 * 4
        rp, cap = r4(n + 1), r4(n + e)
        return [0, rp, 2 * rp, 2 * rp + cap, 2 * rp + 2 * cap, 2 * rp + 3 * cap, 2 * rp + 4 * cap]

    @classmethod
    def _graph_from_block(cls, block, n, e, dev):
        """``CSRGraph`` views of the device copy of a gda_sampler_csr_norm block (one H2D copy)."""
        from .graph import CSRGraph
        cap = n + e
        b = block.to(dev, non_blocking=True)
        o = cls._csr_offsets(n, e)
        sizes = [n + 1, n + 1, cap, cap, cap, cap]
        part = lambda k: b[o[k]:o[k] + max(sizes[k], 1)]
        g = CSRGraph(n, cap, part(0), part(2), part(4).view(torch.float32), part(1), part(3), part(5).view(torch.float32))
        g._nnz = int(block[n])                 # rowptr[n] on the host copy: no device read-back
        g.transient = True
        return g

    def assemble(self, data, seeds, n_id, ei, csr_block=None):
        """Device side of a batch: feature rows by the gather kernel, labels, ids."""
        dev = data.x.device
        if dev.type == "cuda":
            from .ops import gather_rows
            n_dev = n_id.to(dev, non_blocking=True)
            x = gather_rows(data.x, n_dev)
            y = None if data.y is None else data.y[n_dev]
            n_edges = int(ei.size(1))
            ei = ei.to(dev, non_blocking=True)
            ei._gda_trusted = True            # relabelled ids are in range by construction: no validation sync
            if csr_block is not None:         # gcn_norm(edge_index) of this batch, ready made (graph.as_graph)
                ei._gda_prebuilt = self._graph_from_block(csr_block, int(n_id.numel()), n_edges, dev)
            return Data(x=x, edge_index=ei, y=y, n_id=n_dev,
                        batch_size=int(torch.as_tensor(seeds).numel()))
        return Data(x=data.x[n_id], edge_index=ei, y=None if data.y is None else data.y[n_id], n_id=n_id,
                    batch_size=int(torch.as_tensor(seeds).numel()))


import os as _os
# one launch for the K interior steps of a sampled batch (csrc/gda_interior.inc); 0 keeps the K-launch chain
INTERIOR_LDS = _os.environ.get("PYGDA_AMD_INTERIOR_LDS", "1") == "1"


class _PendingBatch:
    """A batch the device sampler has enqueued: capacity-sized device arrays + the event after which the
    counts ({n_nodes, n_edges, nnz, status, n_interior, verdicts of the two interior K-step plans}) are on the host."""
    __slots__ = ("nodes", "ei", "csr", "counts_host", "event", "n_seeds", "stream", "short_rows", "plans", "plan_ok", "T_int")
    recycled = False

    def wait(self):
        self.event.synchronize()
        return self._sizes(self.counts_host.tolist()[:8])

    def _sizes(self, counts):
        n, e, nnz, status, n_int, ok_f, t_f, ok_b = (int(v) for v in counts)
        self.plan_ok = (ok_f > 0, ok_b > 0) if self.plans is not None else (False, False)
        self.T_int = t_f if (self.plans is not None and ok_f > 0) else None      # off-diagonal entries of the interior block
        if status == 2:
            raise _lib.GdaError("gda_dsampler_sample: a seed lies outside [0, num_nodes)")
        if status != 0:
            raise _lib.GdaError("gda_dsampler_sample: a capacity bound was exceeded (internal error)")
        return n, e, nnz, n_int


class _Slot(_PendingBatch):
    """A pending batch whose block belongs to a loader's ring (:class:`_Ring`) and is written again ``depth`` batches
    later: the views, the pointers of the one foreign call that fills it and its two events are made ONCE."""
    __slots__ = ("block", "ring", "done", "free", "freed", "counts_np", "args", "marked", "seeds_pin", "gen")
    recycled = True

    def wait(self):
        _lib.check(_lib.lib().gda_event_synchronize(self.done), "gda_event_synchronize")      # off the interpreter lock
        return self._sizes(self.counts_np[:8])


class _Ring:
    """``depth`` recyclable batch blocks of one loader (one seed count, one fan-out list, one sampler stream).

    Hand-over protocol -- producer thread: batch b goes into slot b % depth; the sampler's stream first waits for the
    slot's ``free`` event when the consumer has recorded one.  Consumer thread (``DeviceNeighborSampler.assemble``):
    orders its stream behind the slot's ``done`` event and, having moved on to this batch, records ``free`` of the
    PREVIOUS slot on its stream -- everything it enqueued for that batch is ahead of that record.  With a queue of
    ``prefetch`` batches between the two, depth >= prefetch + 3 guarantees the record precedes the slot's reuse
    (data.py builds prefetch + 4).  The consumer must not read a batch after it has taken the next one from the same
    loader: the trainers' loops and predict() do not (their outputs and label gathers are new tensors).
    Enforced where it can be: every slot counts its generations and the graph of a batch carries the one it was written
    in -- ``graph.as_graph`` (every conv's entry) raises on a batch whose block has been written again, instead of
    aggregating over another batch's graph.  Stream requirement: ``assemble()`` must run on a stream that is ordered
    behind everything the consumer enqueued for the PREVIOUS batch (the trainers run a whole step on the stream that is
    current when they take the next batch, or join their side streams back into it): the ``free`` record covers that
    stream only."""

    def __init__(self, depth, interior_rows=0):
        self.depth, self.slots, self.key, self.at, self.last = int(depth), None, None, 0, None
        # > 0: every batch declares its first min(interior_rows, n) rows interior (gda_dsampler_batch_ex) -- the static
        # shape the captured sampled step replays at (pygda_amd/sampled_graph.py)
        self.interior_rows = int(interior_rows)

    def reset(self):
        """A new pass over the loader: the caller has ordered the sampler's stream behind the consumer's."""
        self.at, self.last = 0, None
        for s in self.slots or ():
            s.freed = False

    def __del__(self):
        try:
            L = _lib.lib()
            for s in self.slots or ():
                L.gda_event_destroy(s.done)
                L.gda_event_destroy(s.free)
        except Exception:                 # noqa: BLE001 -- interpreter shutdown
            pass


class DeviceNeighborSampler:
    """Python face of the device neighbour sampler (csrc/gda_dsampler.hip): the graph's in-neighbour lists live in
    HBM, a batch (global ids, local edge list, GCN-normalised CSR pair) is built by a few dozen small launches on
    the caller's stream -- the same batches as :class:`NeighborSampler`, bit for bit, with no host work beyond the
    enqueue and one 32-byte read-back of the sizes."""

    MAX_WORKSPACE = int(float(__import__("os").environ.get("PYGDA_AMD_DSAMPLER_MAX_WS_GB", "4")) * 2 ** 30)

    def __init__(self, edge_index, num_nodes):
        _lib.require_gpu_tensor(edge_index, "edge_index", torch.int64)
        L = _lib.lib()
        dev = edge_index.device
        self.device, self.num_nodes, self.num_edges = dev, int(num_nodes), int(edge_index.size(1))
        N, E = self.num_nodes, self.num_edges
        need = L.gda_dsampler_graph_workspace_bytes(E, N)
        if need == 0:
            raise _lib.GdaError("device sampler: graph beyond the int32 index range")
        self.in_ptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
        self.in_src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        status = torch.empty(2, dtype=torch.int32, device=dev)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
        _lib.check(L.gda_dsampler_build_graph(_lib.ptr(src), _lib.ptr(dst), E, N, _lib.ptr(self.in_ptr),
                                              _lib.ptr(self.in_src), _lib.ptr(status), _lib.ptr(ws), ws.numel(),
                                              _lib.stream()), "gda_dsampler_build_graph")
        bad, self.max_in_degree = (int(v) for v in status.tolist())       # one sync per graph
        del ws
        if bad:
            raise IndexError(f"edge_index values must lie in [0, {N}): {bad} edges do not")
        self._ws = {}
        self._layouts = {}
        self._warmed = set()

    def _caps(self, n_seeds, fanouts):
        fan = np.ascontiguousarray(np.asarray(fanouts, dtype=np.int32))
        nc, ec = ctypes.c_int64(), ctypes.c_int64()
        L = _lib.lib()
        st = L.gda_dsampler_caps(int(n_seeds), fan.ctypes.data, fan.size, self.max_in_degree, self.num_edges,
                                 self.num_nodes, ctypes.byref(nc), ctypes.byref(ec))
        if st != 0:
            return None
        need = L.gda_dsampler_workspace_bytes(int(n_seeds), fan.ctypes.data, fan.size, self.max_in_degree,
                                              self.num_edges, self.num_nodes)
        return fan, nc.value, ec.value, need

    def supports(self, n_seeds, fanouts):
        """Fan-outs 1..64 (or -1 when the whole-neighbourhood bound still fits the workspace budget)."""
        caps = self._caps(n_seeds, fanouts)
        return caps is not None and 0 < caps[3] <= self.MAX_WORKSPACE

    def description(self):
        return "device sampler (csrc/gda_dsampler.hip): in-neighbour lists in HBM, no host sampling threads"

    def _layout(self, n_seeds, fanouts, csr, plans):
        """Byte offsets of one batch's device arrays inside ONE block (256-byte aligned each) -- computed once per
        (seed count, fan-outs): a batch then costs the producer thread one allocation and a handful of views instead of
        a dozen allocator calls under the interpreter lock the training thread is waiting for."""
        key = (int(n_seeds), tuple(int(f) for f in fanouts), bool(csr), bool(plans))
        hit = self._layouts.get(key)
        if hit is None:
            caps = self._caps(int(n_seeds), fanouts)
            if caps is None:
                raise _lib.GdaError(f"device sampler: fan-outs {list(fanouts)} are not supported (0, or above 64)")
            fan, ncap, ecap, need = caps
            off, at = {}, 0

            def take(name, nbytes):
                nonlocal at
                off[name] = (at, nbytes)
                at += (nbytes + 255) // 256 * 256

            take("counts", 12 * 8)
            take("seeds", max(int(n_seeds), 1) * 8)
            take("nodes", ncap * 8)
            take("ei", 2 * ecap * 8)
            if csr:
                cap = ecap + ncap
                for name, nbytes in (("rp", (ncap + 1) * 4), ("ci", cap * 4), ("va", cap * 4),
                                     ("trp", (ncap + 1) * 4), ("tci", cap * 4), ("tva", cap * 4)):
                    take(name, nbytes)
            nb = int(_lib.lib().gda_interior_plan_bytes()) if plans else 0
            if plans:
                take("plan0", nb)
                take("plan1", nb)
            hit = self._layouts[key] = (fan, ncap, ecap, need, off, at, nb)
        return hit

    def new_ring(self, depth, interior_rows=0):
        return _Ring(depth, interior_rows)

    def _make_slots(self, ring, key, layout, stream, csr, plans, short_rows, n_seeds):
        """The ring's blocks, views, events and argument lists (once per loader)."""
        fan, ncap, ecap, need, off, total, nb = layout
        L, dev = _lib.lib(), self.device
        ws = self._ws.get(stream.cuda_stream)
        if ws is None or ws.numel() < need:
            ws = self._ws[stream.cuda_stream] = torch.empty(need, dtype=torch.uint8, device=dev)
        # pinned per slot: the 12 counts coming home and the seeds going out (a copy out of pageable memory is staged
        # synchronously by the runtime: the producer thread would sit inside the copy call until its stream -- possibly
        # waiting for the slot's `free` event -- gets there)
        pinned = torch.empty(ring.depth, 12 + max(int(n_seeds), 1), dtype=torch.int64).pin_memory()
        ring._pinned = pinned
        slots = []
        for i in range(ring.depth):
            sl = _Slot()
            # zeroed once: the tail of a capacity-sized array behind a batch's live part then always holds valid node ids
            # (zeros, or an earlier batch's) -- a consumer that runs at capacity shape gathers those rows too
            block = sl.block = torch.zeros(total, dtype=torch.uint8, device=dev)
            base = block.data_ptr()
            at = lambda name: base + off[name][0]
            view = lambda name, dtype: block[off[name][0]:off[name][0] + off[name][1]].view(dtype)
            sl.n_seeds, sl.stream, sl.short_rows, sl.ring, sl.event = int(n_seeds), stream, short_rows, ring, None
            sl.nodes = view("nodes", torch.int64)
            sl.ei = view("ei", torch.int64).view(2, ecap)
            names = ("rp", "ci", "va", "trp", "tci", "tva")
            sl.csr = ((view("rp", torch.int32), view("ci", torch.int32), view("va", torch.float32),
                       view("trp", torch.int32), view("tci", torch.int32), view("tva", torch.float32))
                      if csr else (None,) * 6)
            csr_ptrs = [at(k) for k in names] if csr else [None] * 6
            sl.plans = (view("plan0", torch.uint8), view("plan1", torch.uint8)) if plans else None
            sl.plan_ok = (False, False)
            sl.counts_host = pinned[i, :12]
            sl.counts_np = pinned[i, :12].numpy()
            sl.seeds_pin = pinned[i, 12:]
            ev = [ctypes.c_void_p(), ctypes.c_void_p()]
            for e in ev:
                _lib.check(L.gda_event_create(ctypes.byref(e)), "gda_event_create")
            sl.done, sl.free, sl.freed, sl.marked = ev[0].value, ev[1].value, False, None
            sl.gen = 0                   # bumped every time the block is written again: batches carry the value they saw
            # gda_dsampler_batch_ex's arguments; [5] = seeds, [10] = generator seed, [25] = wait_event change per batch
            sl.args = [_lib.ptr(self.in_ptr), _lib.ptr(self.in_src), self.num_nodes, self.num_edges, self.max_in_degree,
                       None, int(n_seeds), at("seeds"), fan.ctypes.data, fan.size, None,
                       at("nodes"), at("ei"), at("ei") + ecap * 8, *csr_ptrs,
                       at("counts"), at("plan0") if plans else None, at("plan1") if plans else None, nb,
                       pinned[i].data_ptr(), None, sl.done, ring.interior_rows, _lib.ptr(ws), ws.numel(), stream.cuda_stream]
            slots.append(sl)
        ring.slots, ring.key, ring._fan, ring._ws = slots, key, fan, ws
        return slots

    def _enqueue_slot(self, ring, seeds_t, fanouts, seed, csr, plans, short_rows, stream):
        """One foreign call: the next slot of the ring, filled on the sampler's stream (None: not this ring's shape)."""
        key = (stream.cuda_stream, int(seeds_t.numel()), tuple(int(f) for f in fanouts), bool(csr), bool(plans))
        if ring.slots is None:
            self._make_slots(ring, key, self._layout(seeds_t.numel(), fanouts, csr, plans), stream, csr, plans, short_rows,
                             seeds_t.numel())
        if ring.key != key:
            return None
        sl = ring.slots[ring.at % ring.depth]
        ring.at += 1
        a = sl.args
        if seeds_t.is_cuda:
            a[5] = seeds_t.data_ptr()
        else:
            # this slot's pinned seeds are free again: the copy that read them precedes the slot's previous `done` event,
            # which this thread waited for (p.wait) before it enqueued anything else
            sl.seeds_pin[:seeds_t.numel()].copy_(seeds_t)
            a[5] = sl.seeds_pin.data_ptr()
        a[10] = ctypes.c_uint64(int(seed) & (2 ** 64 - 1))
        a[25] = sl.free if sl.freed else None
        sl.freed = False
        sl.gen += 1                      # graphs handed out for the block's previous contents are stale from here on
        _lib.check(_lib.lib().gda_dsampler_batch_ex(*a), "gda_dsampler_batch_ex")
        return sl

    def enqueue(self, seeds, fanouts, seed=0, csr=True, ring=None):
        """Launch the batch on the CURRENT stream; returns a :class:`_PendingBatch` (``ring``: into the next block of a
        loader's :class:`_Ring` with one foreign call, when the batch has the ring's shape)."""
        seeds_t = torch.as_tensor(seeds)
        short_rows = len(fanouts) > 0 and min(int(f) for f in fanouts) > 0    # every row holds at most fan-out + 1 entries
        plans = bool(csr and short_rows and INTERIOR_LDS)
        if ring is not None and seeds_t.dtype == torch.int64 and seeds_t.is_contiguous():
            sl = self._enqueue_slot(ring, seeds_t, fanouts, seed, csr, plans, short_rows, torch.cuda.current_stream())
            if sl is not None:
                return sl
        fan, ncap, ecap, need, off, total, nb = self._layout(seeds_t.numel(), fanouts, csr, plans)
        dev = self.device
        stream = torch.cuda.current_stream()
        ws = self._ws.get(stream.cuda_stream)
        if ws is None or ws.numel() < need:
            ws = self._ws[stream.cuda_stream] = torch.empty(need, dtype=torch.uint8, device=dev)
        seeds_d = seeds_t.to(dev, torch.int64, non_blocking=True).contiguous()
        p = _PendingBatch()
        p.n_seeds, p.stream, p.short_rows = int(seeds_d.numel()), stream, short_rows
        warm = (stream.cuda_stream, total)
        if warm not in self._warmed:
            # the caching allocator keeps its pools per stream and hands a block back only after the streams that used it
            # have passed the point of its release: with the device a step or two behind the host a loader needs more
            # blocks than it holds batches.  Eight of them are put into this stream's pool once, so that no hipMalloc
            # (tens of milliseconds for a block of this size) falls into a training step later
            self._warmed.add(warm)
            spare = [torch.empty(total, dtype=torch.uint8, device=dev) for _ in range(8)]
            del spare
        block = torch.empty(total, dtype=torch.uint8, device=dev)
        base = block.data_ptr()

        def view(name, dtype):
            a, nbytes = off[name]
            return block[a:a + nbytes].view(dtype)

        counts = view("counts", torch.int64)
        counts.zero_()
        p.nodes = view("nodes", torch.int64)
        p.ei = view("ei", torch.int64).view(2, ecap)
        if csr:
            p.csr = (view("rp", torch.int32), view("ci", torch.int32), view("va", torch.float32),
                     view("trp", torch.int32), view("tci", torch.int32), view("tva", torch.float32))
            csr_ptrs = [base + off[k][0] for k in ("rp", "ci", "va", "trp", "tci", "tva")]
        else:
            p.csr = (None,) * 6
            csr_ptrs = [None] * 6
        L = _lib.lib()
        st = _lib.stream()
        _lib.check(L.gda_dsampler_sample(_lib.ptr(self.in_ptr), _lib.ptr(self.in_src), self.num_nodes, self.num_edges,
                                         self.max_in_degree, _lib.ptr(seeds_d), p.n_seeds, fan.ctypes.data, fan.size,
                                         ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), base + off["nodes"][0],
                                         base + off["ei"][0], base + off["ei"][0] + ecap * 8, *csr_ptrs,
                                         base + off["counts"][0], _lib.ptr(ws), ws.numel(), st),
                   "gda_dsampler_sample")
        p.plans, p.plan_ok = None, (False, False)
        if plans:
            # the register programs of the one-launch interior K-step (csrc/gda_interior.inc), one per direction, built
            # HERE -- on the sampler's stream, from the CSR pair that was just built -- so the training stream sees no
            # extra launch; their verdicts ride home with the batch's sizes (counts[5:7], counts[7:9]: {q, T} per direction)
            p.plans = (view("plan0", torch.uint8), view("plan1", torch.uint8))
            cbase = base + off["counts"][0]
            for k in range(2):
                _lib.check(L.gda_interior_plan_build(csr_ptrs[3 * k], csr_ptrs[3 * k + 1], csr_ptrs[3 * k + 2], cbase + 4 * 8,
                                                     base + off[f"plan{k}"][0], nb, cbase + (5 + 2 * k) * 8, st),
                           "gda_interior_plan_build")
        p.counts_host = self._pinned_counts()
        p.counts_host.copy_(counts, non_blocking=True)
        p.event = torch.cuda.Event()
        p.event.record(stream)
        return p

    def _pinned_counts(self):
        """A 12-word pinned landing pad for a batch's counts, from a ring of 64 (a batch's counts are read long before
        the ring comes round; pinning a fresh block per batch costs a host allocation each time)."""
        ring = getattr(self, "_count_ring", None)
        if ring is None:
            ring = self._count_ring = [torch.empty(64, 12, dtype=torch.int64).pin_memory(), 0]
        i = ring[1]
        ring[1] = (i + 1) % 64
        return ring[0][i]

    def sample(self, seeds, fanouts, seed=0):
        """-> (n_id, edge_index) on the device, like :meth:`NeighborSampler.sample` (tests)."""
        p = self.enqueue(seeds, fanouts, seed, csr=False)
        n, e, _, _ = p.wait()
        return p.nodes[:n], p.ei[:, :e]

    def graph_of(self, p, n, e, nnz, n_int=None):
        """:class:`CSRGraph` views of a pending batch's CSR pair (what ``as_graph(edge_index, n)`` would build)."""
        from .graph import CSRGraph
        rp, ci, va, trp, tci, tva = p.csr
        g = CSRGraph(n, n + e, rp[:n + 1], ci, va, trp[:n + 1], tci, tva)
        g._nnz = nnz
        g.transient = True
        if getattr(p, "recycled", False):
            g._slot, g._gen = p, p.gen
        if n_int is not None and p.short_rows:      # rows [n_int, n): the last hop's discoveries, self loop only
            g.n_interior = int(n_int)
            if p.plans is not None:
                g.iplan = (p.plans[0] if p.plan_ok[0] else None, p.plans[1] if p.plan_ok[1] else None)
                g.iplan_T = getattr(p, "T_int", None)
        return g

    def release(self, p):
        """The consumer has ENQUEUED everything that reads batch ``p`` on the current stream and will not look at it again:
        hand its ring block back now (instead of when the next batch is assembled).  For consumers that mix
        :meth:`assemble` with their own reading of the raw slots (the captured sampled step's eager fall-back): without
        it the block would come round unguarded -- ``assemble`` of the NEXT batch is what normally records ``free``."""
        if not getattr(p, "recycled", False):
            return
        _lib.check(_lib.lib().gda_event_record(p.free, torch.cuda.current_stream().cuda_stream), "gda_event_record")
        p.freed = True
        if p.ring.last is p:
            p.ring.last = None

    def assemble(self, data, p, sizes=None):
        """Consumer side (training stream): order behind the sampler's stream, gather the feature rows."""
        n, e, nnz, n_int = sizes if sizes is not None else p.wait()
        cur = torch.cuda.current_stream()
        if p.recycled:
            L, ring, h = _lib.lib(), p.ring, cur.cuda_stream
            _lib.check(L.gda_stream_wait_event(h, p.done), "gda_stream_wait_event")
            prev = ring.last
            if prev is not None and prev is not p:       # this stream has moved on from the previous batch: its block may
                _lib.check(L.gda_event_record(prev.free, h), "gda_event_record")      # be written again behind this point
                prev.freed = True
            ring.last = p
            if p.marked != h:                 # the ring's blocks die with the loader: not before this stream is through
                p.block.record_stream(cur)
                p.marked = h
        elif p.stream != cur:
            cur.wait_event(p.event)
            p.nodes.record_stream(cur)            # ONE block (views share its storage): allocated on the sampler's stream,
                                                  # consumed on this one
        from .ops import gather_rows
        n_id = p.nodes[:n]
        ei = p.ei[:, :e]
        ei._gda_trusted = True
        if p.csr[0] is not None:
            ei._gda_prebuilt = self.graph_of(p, n, e, nnz, n_int)
        x = gather_rows(data.x, n_id)
        y = None if data.y is None else data.y[n_id]
        return Data(x=x, edge_index=ei, y=y, n_id=n_id, batch_size=p.n_seeds)
