"""The SAMPLED mini-batch training step as one hipGraph, replayed at a static shape.

The reference's sampled loop (pygda/models/a2gnn.py:260-277 loaders, :308-319 step loop) meets a new sub-graph every
step: node, edge and interior-row counts all change from batch to batch, so rounds 1-5 issued its ~200 launches per step
eagerly from Python -- on a slow host the step ran at the host's pace (round 5, driver's box: 2.52 ms of host time per
2.53 ms step).  Everything that varies is small, though, and a batch of the device sampler already lives in a
CAPACITY-sized block (sampler._Slot: node ids, edge list, the normalised CSR pair with ``rowptr[i] = nnz`` for every row
behind the live ones, both interior K-step plans).  So the step runs at the capacity shape:

* **rows**: every matrix of the step has ``ncap`` rows (1024 seeds at fan-out [15, 10]: 169,984 against ~157 k live
  ones).  The rows behind the live ones are padding: they gather some valid feature row (the block's tail holds zeros or an
  earlier batch's ids), have no entry in either CSR, are never drawn by the MMD (the draws are made on the host over
  the live count) and are masked out of the cross-entropy on the device (``gda_softmax_nll_*_nv_f32`` reads the live
  count from the batch's own counts) -- so their gradients are exact zeros and nothing they hold reaches a parameter.
* **interior rows**: the loaders' rings are built with ``static_interior`` = the interior capacity (seeds x (1 + f1 + ..)),
  so every batch declares the same number of leading rows interior (gda_dsampler_batch_ex): the one-launch interior
  K-step, its prologue / epilogue and the classifier's interior-rows chain are launched with constants.
* **per step** the host does: wait for the two batches' sizes (the loaders' producer threads have them), order the stream
  behind the sampler's, ONE device-to-device copy per domain of the ring block into the step's static block (~11 MB,
  a few microseconds -- after which the ring block is free again), the MMD draws over the live counts into the static
  sample block (the same CPU-generator draws, in the same order, as the eager ``MMD()``), one ``hipGraphLaunch``.

The capture keeps the eager step's branch structure (source branch and the loss-unused target logits pass on side streams:
the one-launch interior K-step occupies half the chip, the source branch's projections run beside it: 2.13 -> 1.92 ms/step;
`FORK` below), statistics on the main stream.  A batch the static shape cannot take (an
interior plan declined, fewer live rows than twice the interior capacity) runs the ordinary eager step on its real
shape -- same optimiser state, same dropout counters -- and the next one replays again.
"""
import ctypes
import time

import torch

from . import _lib
from .data import Data
from .graph import CSRGraph
from .hipgraph import GraphedStep, _ship
from .utils import mmd as _mmd


import os as _os
ROW_MARGIN = float(_os.environ.get("PYGDA_AMD_SAMPLED_GRAPH_ROW_MARGIN", "0.015"))
# The source branch on a second stream inside the capture (and the loss-unused target logits pass on a third), as the eager
# sampled step runs them.  A forked graph is enqueued node by node (~3 us each) instead of replaying pre-built
# packets -- but these kernels are 30 - 100 us each, the one-launch interior K-step occupies HALF the chip (one workgroup per
# feature column) and the source branch's chip-filling projections fit beside it: measured 2.13 -> 1.92 ms/step at cfg-S
# with 0.37 - 0.42 ms of host work either way.  PYGDA_AMD_SAMPLED_GRAPH_FORK=0: one stream.
FORK = _os.environ.get("PYGDA_AMD_SAMPLED_GRAPH_FORK", "1") == "1"


def interior_capacity(n_seeds, fanouts):
    """Rows a batch can hold before its last hop's discoveries: seeds x (1 + f1 + f1 f2 + ...) over the first L-1 hops."""
    cap, width = int(n_seeds), int(n_seeds)
    for f in list(fanouts)[:-1]:
        if f <= 0:
            return None
        width *= int(f)
        cap += width
    return cap


def static_shape_ok(loader):
    """Can this loader's batches be replayed at one static shape?  Sampled (not full batch), positive fan-outs (every
    row short), an interior capacity the one-launch K-step takes."""
    from .ops import _interior_lds_limits
    fan = list(loader.num_neighbors)
    if loader.full_batch or not fan or min(fan) <= 0 or not loader.recycle or loader.prefetch <= 0:
        return None
    cap = interior_capacity(loader.batch_size, fan)
    if cap is None or cap > _interior_lds_limits()[0]:
        return None
    return cap


class _StaticBatch:
    """The step's own copy of one domain's batch block + the views the trainer reads."""

    def __init__(self, loader, slot, tag, n_first):
        S = loader._sampler
        fan, ncap_block, ecap, need, off, total, nb = S._layout(slot.n_seeds, loader.num_neighbors, True, True)
        # rows the step runs at: the block's capacity counts every sampled neighbour as a new node; on a large graph the
        # batches' live counts sit a few per cent below it and vary by a fraction of a per cent -- so the static shape is
        # the FIRST batch's live count + 1.5 % (rounded up to 256 rows), never beyond the capacity.  A later batch with
        # more live rows than that takes the eager step (GraphedSampledStep.fallbacks).
        ncap = min(ncap_block, (int(n_first * (1.0 + ROW_MARGIN)) + 255) // 256 * 256)
        self.ncap, self.ecap, self.total, self.loader, self.tag = ncap, ecap, total, loader, tag
        self.ncap_block = ncap_block
        self.n_int = min(interior_capacity(slot.n_seeds, loader.num_neighbors), ncap)
        dev = slot.block.device
        block = self.block = torch.zeros(total, dtype=torch.uint8, device=dev)
        view = lambda name, dtype: block[off[name][0]:off[name][0] + off[name][1]].view(dtype)
        self.counts = view("counts", torch.int64)                  # [0] = live rows: the cross-entropy's n_valid
        self.nodes = view("nodes", torch.int64)[:ncap]
        ei = view("ei", torch.int64).view(2, ecap)
        rp, ci, va = view("rp", torch.int32), view("ci", torch.int32), view("va", torch.float32)
        trp, tci, tva = view("trp", torch.int32), view("tci", torch.int32), view("tva", torch.float32)
        g = CSRGraph(ncap, ncap_block + ecap, rp[:ncap + 1], ci, va, trp[:ncap + 1], tci, tva)
        g._nnz = ncap_block + ecap              # bookkeeping only (an upper bound): no kernel takes the entry count as a parameter
        g.transient = True
        g.n_interior = self.n_int
        g.iplan = (view("plan0", torch.uint8), view("plan1", torch.uint8))
        g.tag = tag
        ei._gda_trusted = True
        ei._gda_prebuilt = g
        self.graph, self.ei = g, ei
        self.data = Data(x=None, edge_index=ei, y=None, n_id=self.nodes, batch_size=slot.n_seeds)
        self.live = None                  # (n, e, nnz, n_int) of the batch the block holds now

    def takes(self, slot, sizes):
        n, e, nnz, n_int = sizes
        return (getattr(slot, "recycled", False) and slot.plans is not None and all(slot.plan_ok)
                and n_int == self.n_int and 2 * self.n_int <= n <= self.ncap and slot.block.numel() == self.total)

    def load(self, slot, sizes, stream_handle):
        """Order the stream behind the sampler's, copy the ring block, hand the ring block back."""
        L = _lib.lib()
        _lib.check(L.gda_stream_wait_event(stream_handle, slot.done), "gda_stream_wait_event")
        self.block.copy_(slot.block, non_blocking=True)
        _lib.check(L.gda_event_record(slot.free, stream_handle), "gda_event_record")
        slot.freed = True
        if slot.marked != stream_handle:
            slot.block.record_stream(torch.cuda.current_stream())
            slot.marked = stream_handle
        self.live = sizes

    def materialise(self):
        """Inside the step: the feature rows and labels of the block's node ids (all ``ncap`` of them)."""
        from .ops import GatheredRows, TALL_FUSED, gather_rows
        src = self.loader.data
        d = self.data
        # the feature rows stay where they are: the first projection (and its weight gradient) read them through the ids
        d.x = GatheredRows(src.x, self.nodes) if TALL_FUSED else gather_rows(src.x, self.nodes)
        d.y = None if src.y is None else src.y[self.nodes]
        if d.y is not None:
            d.y._gda_valid_rows = self.counts[0:1]
        return d


class GraphedSampledStep(GraphedStep):
    """``step(slot_s, sizes_s, slot_t, sizes_t)`` -> ticket; ``result(ticket)`` -> (loss, #correct, live source rows).

    ``capture=False`` runs the same static-shape step eagerly every time (tests: captured == eager, bit for bit)."""

    def __init__(self, trainer, net, step_fn, optimizer, source_loader, target_loader, capture=True):
        self.trainer, self.net = trainer, net
        self.loaders = (source_loader, target_loader)
        self.want_capture = capture
        self.static = None
        self.fallbacks = 0                # steps that ran eagerly on their real shape
        self.replays = 0
        self.host_wait_s = 0.0            # of step(): blocked in event waits (the device is behind: back-pressure)
        self.host_work_s = 0.0            # of step(): everything else (draws, counting sorts, copies' enqueue, the launch)
        self._step2 = step_fn             # step_fn(src, tgt, alpha, epoch) of the trainer
        self._pins = None
        super().__init__(lambda s, t: step_fn(s, t, 0.0, 0), optimizer, None, None, warmup=2)

    # -- the MMD draws: over the LIVE row counts, into blocks sized for the capacity --------------------------------------
    def _fill_one(self, key):
        from .ops import selection_csr_host
        ns, nt, times, n = key[:4]
        live_s, live_t = self.static[0].live[0], self.static[1].live[0]
        e = self._samples[key]
        k = e["turn"]
        e["turn"] = 1 - k
        if e["done"][k] is not None:
            w0 = time.perf_counter()
            e["done"][k].synchronize()
            self.host_wait_s += time.perf_counter() - w0
        pins = e["pinv"][k]
        torch.randint(live_s, (times, n), out=pins[0])              # eager MMD()'s draws, same order
        torch.randint(live_t, (times, n), out=pins[1])
        selection_csr_host(pins[0], ns, 0, 2 * n, out=(pins[2], pins[3]))
        selection_csr_host(pins[1], nt, n, 2 * n, out=(pins[4], pins[5]))
        _ship(e["dev"], e["pin"][k])                               # a kernel of this stream reading the pinned block
        ev = torch.cuda.Event()
        ev.record()
        e["done"][k] = ev

    # -- one static-shape step (eager or under capture) -------------------------------------------------------------------
    def _run(self, with_stats=False):
        from .ops import ce_stats_for, dropout_state
        dev = self.static[0].block.device
        bump = getattr(self.optimizer, "bump_steps", None)
        if bump is not None and bump(dropout_state.counter(dev)):
            dropout_state.site = 0
        else:
            dropout_state.next_step(dev)
        src, tgt = self.static[0].materialise(), self.static[1].materialise()
        self.net.train()
        loss, logits = self.step_fn(src, tgt)
        stats = ce_stats_for(logits, src.y)                        # {ce, #correct over the live rows}: by-product of the loss kernel
        if stats is None:
            live = self.static[0].counts[0]
            rows = torch.arange(logits.size(0), device=dev) < live
            correct = ((logits.detach().argmax(dim=1) == src.y) & rows).sum()
            stats = torch.stack([loss.detach().double(), correct.double()])
        else:
            stats[0:1].copy_(loss.detach().reshape(1))             # slot 0: the TOTAL loss of the step
        self.stats = stats
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss, logits

    def _setup(self, slot_s, sizes_s, slot_t, sizes_t):
        """First eligible pair: static blocks, warm-up (rolled back), capture."""
        from .ops import dropout_state
        self.static = (_StaticBatch(self.loaders[0], slot_s, "source", sizes_s[0]),
                       _StaticBatch(self.loaders[1], slot_t, "target", sizes_t[0]))
        if not (self.static[0].takes(slot_s, sizes_s) and self.static[1].takes(slot_t, sizes_t)):
            self.static = None
            return False
        h = torch.cuda.current_stream().cuda_stream
        self.static[0].load(slot_s, sizes_s, h)
        self.static[1].load(slot_t, sizes_t, h)
        self.src, self.tgt = self.static[0].data, self.static[1].data
        self.src.x = self.tgt.x = torch.empty(0, device=self.static[0].block.device)   # (GraphedStep reads .x.device)
        dev = self.static[0].block.device
        prev = _mmd.sample_provider
        _mmd.sample_provider = self._provider
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        saved = [p.detach().clone() for p in params]
        had_state = {id(p): bool(self.optimizer.state.get(p)) for p in params}
        saved_state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v)
                               for k, v in self.optimizer.state[p].items()} for p in params if self.optimizer.state.get(p)}
        cpu_rng = torch.get_rng_state()
        counter = dropout_state.counter(dev)
        saved_counter = counter.clone()
        prev_overlap = getattr(self.trainer, "overlap_sampled", None)
        self.trainer.overlap_sampled = FORK                        # ONE stream by default: no fork inside the capture
        import gc
        torch.cuda.synchronize()
        gc.collect()
        try:
            from . import ops as _ops
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for k in range(self.warmup):
                    self._refill()
                    keep_log, _ops.aggregation_log = _ops.aggregation_log, ([] if k == 0 else _ops.aggregation_log)
                    try:
                        self._run()
                    finally:
                        if k == 0:
                            # which aggregation calls a step makes: (domain, K, path) -- a replayed step logs nothing,
                            # edge counts are formed from this template and the batches' live sizes (edges_of)
                            self.template = [(getattr(e[0], "tag", None), e[1], e[3] if len(e) > 3 else None)
                                             for e in _ops.aggregation_log]
                            _ops.aggregation_log = keep_log
            torch.cuda.current_stream().wait_stream(side)
            if self.want_capture:
                self._refill()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self._run(with_stats=True)
                self._preroll()
            # roll the warm-up back: parameters, optimiser state, the CPU generator, the dropout step counter
            with torch.no_grad():
                for p, v in zip(params, saved):
                    p.copy_(v)
                for p in params:
                    st = self.optimizer.state.get(p)
                    if not st:
                        continue
                    old = saved_state.get(id(p))
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            if old is not None:
                                v.copy_(old[k])
                            else:
                                v.zero_()
                counter.copy_(saved_counter)
            torch.set_rng_state(cpu_rng)
            for e in self._samples.values():                       # the blocks' next refill starts from a clean turn
                for ev in e["done"]:
                    if ev is not None:
                        ev.synchronize()
        finally:
            _mmd.sample_provider = prev
            if prev_overlap is not None:
                self.trainer.overlap_sampled = prev_overlap
        self._pins = [torch.zeros(2, dtype=torch.float64).pin_memory() for _ in range(4)]
        self._pin_events, self._pin_turn = [None] * 4, 0
        return True

    # -- public -------------------------------------------------------------------------------------------------------------
    def step(self, slot_s, sizes_s, slot_t, sizes_t):
        """One training step on the pair of raw batches.  Returns a ticket for :meth:`result`, or None when the pair
        does not fit the static shape (the caller then runs its eager step on the assembled batches)."""
        t0, wait0 = time.perf_counter(), self.host_wait_s
        try:
            return self._step(slot_s, sizes_s, slot_t, sizes_t)
        finally:
            self.host_work_s += (time.perf_counter() - t0) - (self.host_wait_s - wait0)

    def _step(self, slot_s, sizes_s, slot_t, sizes_t):
        if self.static is None:
            if self._setup_failed():
                return None
            ok = self._setup(slot_s, sizes_s, slot_t, sizes_t)
            if not ok:
                self._failed = True
                return None
        elif not (self.static[0].takes(slot_s, sizes_s) and self.static[1].takes(slot_t, sizes_t)):
            self.fallbacks += 1
            return None
        else:
            h = torch.cuda.current_stream().cuda_stream
            self.static[0].load(slot_s, sizes_s, h)
            self.static[1].load(slot_t, sizes_t, h)
        prev = _mmd.sample_provider
        _mmd.sample_provider = self._provider
        prev_overlap = getattr(self.trainer, "overlap_sampled", None)
        try:
            self._refill()
            if self.graph is not None:
                self.graph.replay()
            else:
                self.trainer.overlap_sampled = FORK
                self._run(with_stats=True)
        finally:
            _mmd.sample_provider = prev
            if prev_overlap is not None:
                self.trainer.overlap_sampled = prev_overlap
        self.replays += 1
        k = self._pin_turn
        self._pin_turn = (k + 1) % len(self._pins)
        if self._pin_events[k] is not None:
            w0 = time.perf_counter()
            self._pin_events[k].synchronize()
            self.host_wait_s += time.perf_counter() - w0
        self._pins[k].copy_(self.stats, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pin_events[k] = ev
        return (k, sizes_s[0])

    def edges_of(self, sizes_s, t_s, sizes_t, t_t):
        """``(reference-equivalent, executed)`` edge aggregations of ONE static-shape step on batches of these live sizes
        (``sizes`` = (n, e, nnz, n_interior), ``t`` = off-diagonal entries of the interior block): SURVEY 8(d)'s count --
        every call = K full aggregations of the batch's nnz -- and the entries whose multiply-add the step executes
        (ops._launch_kstep_interior's accounting on the live counts)."""
        live = {"source": (sizes_s, t_s), "target": (sizes_t, t_t)}
        ref = done = 0
        for tag, K, kind in self.template:
            (n, _, nnz, n_int), T = live[tag]
            ref += K * nnz
            if kind == "interior-lds" and T is not None:
                done += nnz + (K - 1) * (T + n_int)
            elif kind is not None:
                done += K * (nnz - (n - n_int)) + (n - n_int)
            else:
                done += K * nnz
        return ref, done

    def _setup_failed(self):
        return getattr(self, "_failed", False)

    def result(self, ticket):
        """(loss, #correct source rows, live source rows) of the step behind ``ticket`` (at most three newer steps)."""
        k, n = ticket
        self._pin_events[k].synchronize()
        loss, correct = self._pins[k].tolist()
        return loss, correct, n
