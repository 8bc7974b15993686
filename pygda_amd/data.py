"""Minimal graph container and loaders with the PyG names pygda's trainers use
(``Data``; ``NeighborLoader(data, num_neighbors, batch_size=...)``;
pygda/models/a2gnn.py:254-286).  Full-batch loading hands out the graph itself (cached on
the device after the first ``.to``); fan-out sampling is done by the native host sampler
(pygda_amd/sampler.py)."""
import torch
import torch.utils.data


class Data:
    """Any object with ``x [N,F] fp32``, ``edge_index [2,E] int64`` (row 0 = source, row 1 =
    destination), ``y [N] int64`` and ``.to(device)`` satisfies the trainers; this is one."""

    def __init__(self, x=None, edge_index=None, y=None, **kwargs):
        self.x, self.edge_index, self.y = x, edge_index, y
        for k, v in kwargs.items():
            setattr(self, k, v)
        self._device_copies = {}

    # -- container protocol -------------------------------------------------------
    def keys(self):
        return [k for k, v in self.__dict__.items() if not k.startswith("_") and v is not None]

    def __contains__(self, key):
        return key in self.keys()

    def __getitem__(self, key):
        return getattr(self, key)

    @property
    def num_nodes(self):
        if self.x is not None:
            return self.x.size(0)
        return int(self.edge_index.max()) + 1 if self.edge_index is not None and self.edge_index.numel() else 0

    @property
    def num_edges(self):
        return 0 if self.edge_index is None else self.edge_index.size(1)

    @property
    def num_node_features(self):
        return 0 if self.x is None else self.x.size(1)

    def is_undirected(self):
        ei = self.edge_index
        n = self.num_nodes
        a = torch.unique(ei[0] * n + ei[1])
        b = torch.unique(ei[1] * n + ei[0])
        return a.numel() == b.numel() and bool((a == b).all())

    def to(self, device, non_blocking=False):
        """Device copy, made once per device and then reused -- so per-step
        ``batch.to(device)`` (a2gnn.py:311-312) costs nothing after the first call and the
        graph cache sees the same edge tensor every step."""
        device = torch.device(device)
        probe = self.x if self.x is not None else self.edge_index
        if probe is not None and probe.device == device:
            return self
        key = str(device)
        hit = self._device_copies.get(key)
        if hit is None:
            hit = Data(**{k: (v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
                          for k, v in self.__dict__.items() if not k.startswith("_")})
            self._device_copies[key] = hit
            if device.type == "cuda" and hit.x is not None:
                from . import sparse_features          # bag-of-words inputs: CSR once, SpMM projection
                sparse_features.maybe_register(hit.x)
        if getattr(self, "_static_graph", False) and hit.edge_index is not None:
            hit.edge_index._gda_static = True
        return hit

    def cpu(self):
        return self.to("cpu")

    def __repr__(self):
        info = ", ".join(f"{k}={list(v.shape) if torch.is_tensor(v) else v}" for k, v in self.__dict__.items()
                         if not k.startswith("_") and v is not None)
        return f"Data({info})"


def degree_order(edge_index, num_nodes):
    """``new_id [N]``: the relabelling that numbers the nodes by decreasing in-degree (ties in id order).  On a power-law
    graph the rows most rows gather then share cache lines and pages: the full-graph aggregation of an R-MAT 2^22 graph
    runs 6.8 -> 5.2 ms with it (profiles/HISTORY.md 5, `roofline_hbm_regime.rmat_2^22`); a uniform graph has nothing to gain."""
    deg = torch.bincount(edge_index[1], minlength=num_nodes)
    order = torch.argsort(deg, descending=True, stable=True)          # order[k] = old id of the k-th node
    new_id = torch.empty_like(order)
    new_id[order] = torch.arange(num_nodes, device=order.device)
    return new_id


def relabel(data, new_id):
    """The same graph with node ``i`` renamed ``new_id[i]`` (features, labels, every per-node attribute and the edge
    list follow): an isomorphic ``Data``, so training on it is training on the original; ``out[new_id]`` maps a
    per-node result of the relabelled graph back to the original numbering."""
    n = data.num_nodes
    e = data.num_edges
    inv = torch.empty_like(new_id)
    inv[new_id] = torch.arange(n, device=new_id.device)
    out, private = {}, {}
    for k, v in data.__dict__.items():
        if v is None:
            continue
        if k.startswith("_"):
            # private flags travel (`_static_graph`: the K-step kernel's eligibility); the device-copy cache does not --
            # it holds copies of the OLD numbering
            if k != "_device_copies":
                private[k] = v
        elif k == "edge_index":
            out[k] = new_id[v]
        elif torch.is_tensor(v) and v.dim() >= 1 and v.size(0) == n and not _edge_level(k, v, n, e):
            out[k] = v[inv]
        else:
            out[k] = v                                  # per-edge attributes keep their order: the edges did not move
    res = Data(**out)
    for k, v in private.items():
        setattr(res, k, v)
    return res


def _edge_level(key, v, n, e):
    """Is ``v`` (first dimension n) a per-EDGE attribute?  Only ambiguous when the graph has as many edges as nodes:
    then the PyG naming convention decides (``edge_attr``, ``edge_weight``, ``edge_*`` are per edge; ``x``, ``y``,
    ``batch``, ``pos``, ``*_mask`` per node), and anything else is refused rather than guessed."""
    if e != n:
        return False
    if key.startswith("edge"):
        return True
    if key in ("x", "y", "batch", "pos") or key.endswith("_mask") or key.startswith("node"):
        return False
    raise ValueError(f"relabel: attribute {key!r} has {n} rows and the graph has {n} nodes AND {e} edges -- "
                     "name it edge_* or node_* so that it is permuted (or not) on purpose")


import os as _os
AUTO_REORDER = _os.environ.get("PYGDA_AMD_AUTO_REORDER", "1") == "1"
RECYCLE = _os.environ.get("PYGDA_AMD_LOADER_RECYCLE", "1") == "1"      # NeighborLoader(recycle=True) is honoured
AUTO_REORDER_MIN_NODES = int(_os.environ.get("PYGDA_AMD_AUTO_REORDER_MIN_NODES", str(1 << 20)))
AUTO_REORDER_SKEW = float(_os.environ.get("PYGDA_AMD_AUTO_REORDER_SKEW", "64"))


def auto_reorder(data):
    """``(data', new_id)``: the degree-ordered relabelling of a STATIC full-batch graph when it pays -- at least
    ``AUTO_REORDER_MIN_NODES`` nodes (2^20: below that the feature matrix sits in the Infinity Cache anyway) and a
    largest in-degree of at least ``AUTO_REORDER_SKEW`` x the mean (a uniform graph has nothing to reorder: 9.03 ms
    either way at 5 M nodes) -- else ``(data, None)``.  One bincount + one sort + one row permutation per fit()."""
    ei = data.edge_index
    n = data.num_nodes
    if (not AUTO_REORDER or ei is None or data.x is None or n < AUTO_REORDER_MIN_NODES or ei.numel() == 0
            or getattr(data, "batch", None) is not None):
        return data, None
    deg = torch.bincount(ei[1], minlength=n)
    if float(deg.max()) < AUTO_REORDER_SKEW * max(float(ei.size(1)) / n, 1.0):
        return data, None
    order = torch.argsort(deg, descending=True, stable=True)
    new_id = torch.empty_like(order)
    new_id[order] = torch.arange(n, device=order.device)
    return relabel(data, new_id), new_id


def to_undirected(edge_index, num_nodes=None):
    """Symmetrise and de-duplicate an edge list (what benchmark/node/a2gnn.py:92-97 applies)."""
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    key = torch.unique(both[0] * n + both[1])
    return torch.stack([key // n, key % n])


class NeighborLoader:
    """``NeighborLoader(data, num_neighbors, batch_size)`` as the trainers build it.

    ``num_neighbors`` all -1 with ``batch_size >= N``: one batch, the whole graph (every
    benchmark setting of the reference).  Otherwise seeds are taken ``batch_size`` at a time
    (in order unless ``shuffle``) and each batch is the union of their sampled L-hop
    in-neighbourhoods, seeds first, exactly ``batch.batch_size`` of them."""

    def __init__(self, data, num_neighbors, batch_size=1, shuffle=False, input_nodes=None,
                 rank=0, world_size=1, seed=0, device=None, prefetch=2, full_batch=None, auto_reorder=False, recycle=False,
                 **kwargs):
        self.prefetch = int(prefetch)
        # sampled batches in a ring of prefetch + 4 blocks written round and round (sampler._Ring: one foreign call per batch
        # in the producer thread).  Only for consumers that are done with a batch when they take the next one
        self.recycle = bool(recycle) and RECYCLE
        self._ring = None
        self.num_neighbors = list(num_neighbors)
        self.batch_size, self.shuffle = int(batch_size), shuffle
        self.rank, self.world_size, self.seed = rank, world_size, seed
        n = data.num_nodes
        # ``full_batch=False`` sends a whole-graph request (fan-out -1, one batch) through the sampler
        # anyway: the sampled-batch code path on an input whose result is known (tests)
        full = (all(k == -1 for k in self.num_neighbors) and self.batch_size >= n
                and input_nodes is None and world_size == 1 and full_batch is not False)
        # sampled mode keeps the feature matrix resident on the training device: batches are
        # assembled there by the row-gather kernel, only node / edge ids cross PCIe
        self.data = data.to(device) if (device is not None and not full) else data
        # full batch on a large power-law graph: train on the degree-ordered relabelling (hubs first: the rows most
        # rows gather share cache lines and pages -- 7.85 -> 5.16 ms per aggregation on R-MAT 2^22, profiles/HISTORY.md 5); node i
        # of the caller's numbering is row new_id[i] of every batch, predict() maps results back
        # (``auto_reorder=True``: the trainers whose step reads nothing but the loader's batches ask for it -- a caller
        # that indexes a batch with structures of its own built from the original numbering must not)
        self.new_id = None
        if full and auto_reorder:
            self.data, self.new_id = globals()["auto_reorder"](data)
        self.input_nodes = torch.arange(n) if input_nodes is None else torch.as_tensor(input_nodes).long().cpu()
        self.full_batch = full
        self._sampler = None
        self._epoch = 0

    def _batches(self, epoch=None):
        seeds = self.input_nodes
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + (self._epoch if epoch is None else epoch))
            seeds = seeds[torch.randperm(seeds.numel(), generator=g)]
        chunks = list(torch.split(seeds, self.batch_size))
        # data-parallel shard of the seed batches; every rank gets the same count (short ranks
        # wrap around) so that the per-step gradient all-reduce never waits for a missing peer
        per_rank = -(-len(chunks) // self.world_size)
        return [chunks[(self.rank + i * self.world_size) % len(chunks)] for i in range(per_rank)]

    def __len__(self):
        return 1 if self.full_batch else len(self._batches())

    def sampler_description(self):
        """Which sampler assembles the batches (bench.py: config.sampler)."""
        if self.full_batch:
            return "none (full batch)"
        if self._sampler is None:
            return "not started"
        return self._sampler.description()

    def __iter__(self):
        if self.full_batch:
            self.data._static_graph = True        # the same graph every step: operators derived from it
            if self.data.edge_index is not None:  # (A*A for K-step propagation) may be cached
                self.data.edge_index._gda_static = True
            yield self.data
            return
        from .sampler import NeighborSampler, DeviceNeighborSampler
        if self._sampler is None:
            self._sampler = self._make_sampler()
        # the epoch counter advances when iteration STARTS: ``zip(source_loader, target_loader)`` never
        # resumes the second generator after the first one is exhausted, so a bump after the last yield
        # would leave the target loader on epoch 0 (same neighbourhoods and shuffle every epoch)
        epoch = self._epoch
        self._epoch += 1
        batches = self._batches(epoch)
        seeds_of = lambda b: hash((self.seed, epoch, b, self.rank)) & 0x7FFFFFFF
        if isinstance(self._sampler, DeviceNeighborSampler):
            yield from self._device_batches(batches, seeds_of)
        elif self.prefetch <= 0:
            for b, seeds in enumerate(batches):
                yield self._sampler.sample_batch(self.data, seeds, self.num_neighbors, seed=seeds_of(b))
        else:
            # host sampling of the next batches runs in a background thread (the native sampler
            # releases the GIL) while the device works on the current one; device-side assembly
            # (row gather on the training stream) stays in the consumer thread
            import queue
            import threading
            q, stop = queue.Queue(maxsize=self.prefetch), threading.Event()

            def producer():
                try:
                    for b, seeds in enumerate(batches):
                        if stop.is_set():
                            return
                        q.put((seeds, self._sampler.sample(seeds, self.num_neighbors, seed=seeds_of(b),
                                                           csr=self._sampler.emit_csr(self.data))))
                    q.put(None)
                except BaseException as exc:        # surface sampler errors in the consumer
                    q.put(exc)

            th = threading.Thread(target=producer, daemon=True)
            th.start()
            try:
                while True:
                    item = q.get()
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    seeds, parts = item
                    yield self._sampler.assemble(self.data, seeds, *parts)
            finally:
                stop.set()
                while th.is_alive():                 # unblock a producer waiting on a full queue
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        th.join(timeout=0.01)


    def iter_raw(self):
        """The batches of :meth:`__iter__` WITHOUT their consumer-side assembly: ``(slot, (n, e, nnz, n_interior))`` per
        batch, ``slot`` = the ring block the device sampler filled (sampler._Slot: node ids, edge list, CSR pair, interior
        K-step plans, all capacity-sized; its ``done`` event orders a consumer behind the sampler's stream).  For the
        captured sampled step (pygda_amd/sampled_graph.py), which copies the block into its static buffers and gathers
        the feature rows inside the graph.  Device sampler + recycling ring only (None otherwise)."""
        from .sampler import DeviceNeighborSampler
        if self.full_batch or not self.recycle or self.prefetch <= 0:
            return None
        if self._sampler is None:
            self._sampler = self._make_sampler()
        if not isinstance(self._sampler, DeviceNeighborSampler):
            return None
        epoch = self._epoch
        self._epoch += 1
        batches = self._batches(epoch)
        seeds_of = lambda b: hash((self.seed, epoch, b, self.rank)) & 0x7FFFFFFF
        return self._device_batches(batches, seeds_of, raw=True)

    def _make_sampler(self):
        """The device sampler when the graph and the features live on the GPU and the fan-outs allow it (1..64, or
        -1 within its workspace budget); the native host sampler otherwise (``PYGDA_AMD_DEVICE_SAMPLER=0`` forces it)."""
        import os
        from .sampler import NeighborSampler, DeviceNeighborSampler
        ei = self.data.edge_index
        if (ei.is_cuda and self.data.x is not None and self.data.x.is_cuda
                and os.environ.get("PYGDA_AMD_DEVICE_SAMPLER", "1") == "1"):
            ds = DeviceNeighborSampler(ei, self.data.num_nodes)
            if ds.supports(self.batch_size, self.num_neighbors):
                return ds
        return NeighborSampler(ei, self.data.num_nodes)

    def _device_batches(self, batches, seeds_of, raw=False):
        """Batches from the device sampler.  With prefetching, a producer thread enqueues batch b+1.. on a side
        stream and waits for their sizes there, so the training stream never waits for a size read-back; the
        consumer orders itself behind the side stream with an event and gathers the feature rows."""
        S = self._sampler
        if self.prefetch <= 0:
            for b, seeds in enumerate(batches):
                yield S.assemble(self.data, S.enqueue(seeds, self.num_neighbors, seed=seeds_of(b)))
            return
        import queue
        import threading
        import time
        self.producer_cpu_s, self.producer_batches, self.producer_enqueue_cpu_s = 0.0, 0, 0.0
        dev = self.data.x.device
        if getattr(self, "_samp_stream", None) is None:
            import os
            # the sampler's ~45 small dependent launches per batch sit beside chip-filling training kernels: at the
            # default priority each of them queues behind whatever the training streams have in flight
            self._samp_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("PYGDA_AMD_SAMPLER_PRIORITY", "0")))
        side = self._samp_stream
        side.wait_stream(torch.cuda.current_stream())       # the graph / features may have just been produced
        ring = None
        if self.recycle and not getattr(self, "_ring_busy", False):     # (a second live iterator over this loader allocates)
            if self._ring is None:
                self._ring = S.new_ring(self.prefetch + 4, getattr(self, "static_interior", 0))
            ring = self._ring
            ring.reset()          # (the wait above also puts the last pass's batches behind the sampler's stream)
            self._ring_busy = True
        q, stop = queue.Queue(maxsize=self.prefetch), threading.Event()

        def producer():
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(side):
                    for b, seeds in enumerate(batches):
                        if stop.is_set():
                            return
                        c0 = time.thread_time()
                        p = S.enqueue(seeds, self.num_neighbors, seed=seeds_of(b), ring=ring)
                        self.producer_enqueue_cpu_s += time.thread_time() - c0
                        sizes = p.wait()
                        # on-core time of this thread per batch: what the training thread may have to wait for when it
                        # wants the interpreter lock back (bench.py: config.producer_cpu_ms_per_batch)
                        self.producer_cpu_s += time.thread_time() - c0
                        self.producer_batches += 1
                        q.put((p, sizes))
                q.put(None)
            except BaseException as exc:
                q.put(exc)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item if raw else S.assemble(self.data, *item)
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.01)
            if ring is not None:
                self._ring_busy = False


def collate_graphs(graphs):
    """PyG's ``Batch.from_data_list`` for what pygda's graph mode reads: ``x`` / ``y`` concatenated in list order,
    ``edge_index`` with every graph's node ids shifted by the nodes before it, ``batch`` = graph index per node
    (sorted by construction), ``num_graphs``."""
    counts = torch.tensor([g.x.size(0) for g in graphs], dtype=torch.long)
    offs = (torch.cumsum(counts, 0) - counts).tolist()
    ei = [g.edge_index + o for g, o in zip(graphs, offs)]
    b = Data(x=torch.cat([g.x for g in graphs], dim=0),
             edge_index=torch.cat(ei, dim=1) if ei else torch.empty(2, 0, dtype=torch.long),
             y=torch.cat([g.y.reshape(-1) for g in graphs], dim=0),
             batch=torch.repeat_interleave(torch.arange(len(graphs)), counts), num_graphs=len(graphs))
    return b


class _GraphBatch(Data):
    """A collated batch: ``.to(device)`` tags the device copy of ``batch`` as sorted and with its graph count, so
    the readout needs no device check and no read-back."""

    def to(self, device, non_blocking=False):
        hit = super().to(device, non_blocking)
        if hit.batch is not None:
            hit.batch._gda_sorted = True
            hit.batch._gda_num_graphs = self.num_graphs
        return hit


def _collate(graphs):
    b = collate_graphs(graphs)
    out = _GraphBatch(**{k: getattr(b, k) for k in ("x", "edge_index", "y", "batch", "num_graphs")})
    out.batch._gda_sorted = True
    out.batch._gda_num_graphs = out.num_graphs
    return out


class DataLoader(torch.utils.data.DataLoader):
    """``torch_geometric.loader.DataLoader(dataset, batch_size, shuffle)`` as a2gnn.py:278-286 builds it: torch's own
    loader (its shuffling draws -- the iterator's base seed, then the sampler's seed -- come from the default CPU
    generator exactly as with PyG, whose DataLoader is the same class) with the collation above.  ``dataset``: any
    sequence of ``Data``-like graphs (``x``, ``edge_index``, ``y`` with one label per graph)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        kwargs.pop("collate_fn", None)
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=_collate, **kwargs)
