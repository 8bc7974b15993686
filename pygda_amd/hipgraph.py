"""Whole-step hipGraph capture for full-batch training.

On the citation graphs a training step is ~250 kernels of a few microseconds each: eager
execution is bound by the host's launch rate, not by the GPU (MI355X_MICROARCH.md: eager
goes host-bound below ~3 us per kernel).  With full-batch loading every shape is static, so
forward + backward + Adam are captured once into a hipGraph (``torch.cuda.CUDAGraph`` on
ROCm) and replayed per step.  Our C-ABI kernels are launched on the stream handed to them,
keep no state and never allocate, so they capture like any other kernel.

What changes from step to step enters through static device buffers refilled before each
replay: the MMD row samples (drawn from the CPU generator exactly as in eager mode, so a
seeded run still samples the same rows).  Dropout masks come from torch's graph-safe Philox
state.  Steps whose arithmetic depends on a per-epoch Python scalar (the GRL alpha of the
adversarial branch) are not captured.
"""
import ctypes

import torch

from .utils import mmd as _mmd


# Host-generator draws inside a training step (the reference draws e.g. the gradient-penalty
# interpolation weights with ``torch.rand(...).to(device)``, pygda/models/adagcn.py:423-434).  Eager:
# exactly that.  Captured: a static device buffer per call site, refilled from the same CPU draws (same
# order) before every replay.
host_rand_provider = None


def host_rand(shape, device):
    if host_rand_provider is not None:
        return host_rand_provider(tuple(shape))
    return torch.rand(shape).to(device)


def _col_major(p):
    return p.dim() == 2 and not p.is_contiguous() and p.t().is_contiguous()


def _mem_flat(g, p):
    """``g`` (a gradient of ``p``) flattened in the memory order of ``p``."""
    return g.t().reshape(-1) if _col_major(p) else g.reshape(-1)


def _mem_view(flat, p):
    """The inverse: a view of ``flat`` with the shape AND strides of ``p``."""
    return flat.view(p.size(1), p.size(0)).t() if _col_major(p) else flat.view_as(p)


H2D_KERNEL = __import__("os").environ.get("PYGDA_AMD_H2D_KERNEL", "1") == "1"


def _ship(dst, pinned):
    """``dst.copy_(pinned, non_blocking=True)`` for a small pinned block in front of a replay -- as a kernel of the current
    stream that reads the pinned memory itself (gda_copy_from_pinned) when it can, else the runtime's asynchronous copy."""
    if H2D_KERNEL and dst.is_cuda and dst.is_contiguous() and pinned.is_contiguous():
        from . import _lib
        nbytes = dst.numel() * dst.element_size()
        if nbytes == pinned.numel() * pinned.element_size() and \
                _lib.lib().gda_copy_from_pinned(_lib.ptr(dst), ctypes.c_void_p(pinned.data_ptr()), nbytes, _lib.stream()) == 0:
            return
    dst.copy_(pinned, non_blocking=True)


class _RandSlot:
    def __init__(self, shape, dev):
        self.shape = shape
        self.dev = torch.zeros(shape, dtype=torch.float32, device=dev)
        self.pin = [torch.zeros(shape, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.done, self.turn = [None, None], 0
        self.arena = None          # GraphedStep._consolidate_rand: every slot of a step in ONE block, shipped by ONE copy

    def fill(self):
        if self.arena is not None:                       # the draw now, in call order; the copy once, behind the last fill
            torch.rand(self.shape, out=self.pin[self.arena["half"]])
            self.arena["dirty"] = True
            return
        k = self.turn
        self.turn = 1 - k
        if self.done[k] is not None:
            self.done[k].synchronize()
        torch.rand(self.shape, out=self.pin[k])
        self.dev.copy_(self.pin[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[k] = ev


class _DPSamples:
    """Static device block + double-buffered pinned blocks holding one rank's MMD row samples of both
    domains and the CSRs of their 0/1 selection matrices (backward scatter): drawn on the host from
    the CPU generator in the eager data-parallel MMD()'s order, shipped with one copy."""

    def __init__(self, dev, ns, nt, times, per):
        self.ns, self.nt, self.times, self.per = ns, nt, times, per
        shapes = [((times, per), torch.int64), ((times, per), torch.int64), ((ns + 1,), torch.int32),
                  ((times * per,), torch.int32), ((nt + 1,), torch.int32), ((times * per,), torch.int32)]
        total = sum((torch.empty(0, dtype=dt).element_size() * int(torch.tensor(sh).prod()) + 15) // 16 * 16
                    for sh, dt in shapes)
        self.dev = torch.zeros(total, dtype=torch.uint8, device=dev)
        self.devv = GraphedStep._carve(self.dev, shapes)
        self.pin = [torch.zeros(total, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.pinv = [GraphedStep._carve(b, shapes) for b in self.pin]
        self.done, self.turn = [None, None], 0
        self.ones = torch.ones(times * per, dtype=torch.float32, device=dev)

    def fill(self):
        from .ops import selection_csr_host
        k = self.turn
        self.turn = 1 - k
        if self.done[k] is not None:
            self.done[k].synchronize()
        p = self.pinv[k]
        torch.randint(self.ns, (self.times, self.per), out=p[0])
        torch.randint(self.nt, (self.times, self.per), out=p[1])
        _mmd.apply_row_maps(p[0], p[1], self.ns, self.nt)
        selection_csr_host(p[0], self.ns, 0, self.per, out=(p[2], p[3]))
        selection_csr_host(p[1], self.nt, 0, self.per, out=(p[4], p[5]))
        self.dev.copy_(self.pin[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[k] = ev

    def views(self):
        d = self.devv
        return d[0], d[1], (d[2], d[3], self.ones), (d[4], d[5], self.ones)


class LossTerms(tuple):
    """The step's loss as its independent terms (each a 0-dim tensor with its own autograd history) instead of their
    sum.  A captured step hands this to :class:`GraphedStep`, which seeds every term with a constant 1 in ONE engine
    run and forms the reported total on its statistics branch: the addition of the terms, the fill of the backward
    seed and -- with them -- the loss value's own reduction kernels leave the chain between the domain loss's forward
    and backward kernels (``loss = ce + mmd; loss.backward()`` put three glue launches there)."""


defer_total = False        # set by GraphedStep around its step function: trainers may then return LossTerms
BUMP_AT_START = __import__("os").environ.get("PYGDA_AMD_BUMP_AT_START", "1") == "1"


class GraphedStep:
    """Captures ``loss, logits = step_fn(src, tgt)``, ``backward`` and ``optimizer.step()``."""

    def __init__(self, step_fn, optimizer, src, tgt, warmup=3, dp=False, extra_optimizers=(), unroll=1,
                 inline_stats=False):
        """``dp``: data-parallel step with the library-owned RCCL communicator -- the gradient
        all-reduce and the MMD row all-gather are enqueued on the capturing stream like any kernel,
        so the whole step is still ONE graph."""
        self.step_fn, self.optimizer, self.src, self.tgt, self.dp = step_fn, optimizer, src, tgt, dp
        # A capture that never forks replays through pre-built packets (~0.4 us between dependent kernels, a launch call of
        # ~0.3 us per node); ONE fork anywhere makes the runtime enqueue the graph node by node (3 us of host time and 5 - 6
        # us of device spacing per kernel: profiles/HISTORY.md 4.7).  A trainer whose step is one chain of kernels (GRADE,
        # UDAGCN, AdaGCN, ...: everything but A2GNN's three-branch step) therefore takes its two log numbers on the main
        # stream instead of on the statistics side branch -- two tiny kernels in line against every kernel of the step
        # paying for the fork.  AdaGCN's launch call took 2.6 ms for a 2.6 ms replay: the host never got ahead.
        self.inline_stats = bool(inline_stats)
        self.extra_optimizers = list(extra_optimizers)    # stepped INSIDE step_fn (critics): rolled back too
        self._rand_slots, self._rand_cursor = [], 0
        self._dp_idx = {}                 # (ns, nt, times, per) -> (dev_s, dev_t, pin_s, pin_t)
        self._samples = {}                # (ns, nt, times, n) -> (dev_s, dev_t, pin_s, pin_t)
        self._order = []
        # `unroll` consecutive steps in ONE capture (each with its own sample block): between the last kernel of a
        # replay and the first of the next the device idles 40-50 us on this runtime whatever the host does
        # (profiles/HISTORY.md 4.7); a replay of U steps pays that once.  A one-step graph of the same step serves the remainder.
        self.unroll = max(1, int(unroll)) if (type(self) is GraphedStep and not dp) else 1
        self._sub = 0                     # sub-step being issued (selects the sample block)
        self._fills = {}                  # sub-step -> refill callables, in call order
        self._fill_keys = {}              # sub-step -> the sample blocks' keys, in call order
        self._groups = {}                 # (ns, nt, times, n) -> the sub-steps' blocks as ONE device / pinned allocation
        self.graph = None
        self.warmup = warmup
        self.loss = self.logits = None

    # -- MMD sample plumbing ---------------------------------------------------------
    @staticmethod
    def _carve(block, shapes):
        """Typed views of one byte block (offsets 16-byte aligned)."""
        views, off = [], 0
        for sh, dt in shapes:
            n = 1
            for v in sh:
                n *= v
            nbytes = n * torch.empty(0, dtype=dt).element_size()
            views.append(block[off:off + nbytes].view(dt).view(sh))
            off += (nbytes + 15) // 16 * 16
        return views

    def _provider(self, ns, nt, times, n):
        """Static device buffers: the row samples and the selection CSRs of their scatter -- ONE
        device block and two pinned host blocks (double-buffered), so a refill is one H2D copy and
        the host may prepare step e+1 while the copy of step e is still in flight."""
        key = (ns, nt, times, n, self._sub)
        if key not in self._samples:
            dev = self.src.x.device
            shapes = [((times, n), torch.int64), ((times, n), torch.int64), ((ns + 1,), torch.int32),
                      ((times * n,), torch.int32), ((nt + 1,), torch.int32), ((times * n,), torch.int32)]
            total = sum((torch.empty(0, dtype=dt).element_size() * int(torch.tensor(sh).prod()) + 15) // 16 * 16
                        for sh, dt in shapes)
            group = None
            if self.unroll > 1:
                # The sub-steps of a multi-step capture share ONE device block and one pair of pinned blocks: a replay's
                # refill is then ONE host-to-device copy, whatever the number of steps in the capture.  With a copy per
                # sub-step, four or eight asynchronous copies queued up behind the replay in flight, and once per process
                # hipMemcpyAsync blocked for 5 - 6 ms in one of them (the runtime brings up another copy queue the first
                # time it finds the one in use busy: tools/replay_series.py, profiles/r6_stall_refill.txt) -- the "one slow
                # replay early in a graph's life", inside a 20-step timed region as soon as a capture held four steps.
                total = (total + 255) // 256 * 256
                group = self._groups.get(key[:4])
                if group is None:
                    group = self._groups[key[:4]] = dict(
                        dev=torch.zeros(self.unroll * total, dtype=torch.uint8, device=dev),
                        pin=[torch.zeros(self.unroll * total, dtype=torch.uint8).pin_memory() for _ in range(2)],
                        done=[None, None], turn=0)
                lo = self._sub * total
                dev_block = group["dev"][lo:lo + total]
                pin_blocks = [b[lo:lo + total] for b in group["pin"]]
            else:
                dev_block = torch.zeros(total, dtype=torch.uint8, device=dev)
                pin_blocks = [torch.zeros(total, dtype=torch.uint8).pin_memory() for _ in range(2)]
            ones = torch.ones(times * n, dtype=torch.float32, device=dev)
            self._samples[key] = dict(dev=dev_block, devv=self._carve(dev_block, shapes), pin=pin_blocks,
                                      pinv=[self._carve(b, shapes) for b in pin_blocks],
                                      done=[None, None], turn=0, ones=ones, group=group)
            fill = lambda key=key: self._fill_one(key)      # noqa: E731
            self._order.append(fill)
            self._fills.setdefault(self._sub, []).append(fill)
            self._fill_keys.setdefault(self._sub, []).append(key)
            self._fill_one(key)
        e = self._samples[key]
        d = e["devv"]
        return d[0], d[1], (d[2], d[3], d[4], d[5], e["ones"])

    def _draw_one(self, key, k):
        """The draws of one MMD call of one (sub-)step into half ``k`` of its pinned block."""
        from .ops import selection_csr_host
        ns, nt, times, n = key[:4]
        pins = self._samples[key]["pinv"][k]
        torch.randint(ns, (times, n), out=pins[0])                  # eager MMD()'s draws, same order,
        torch.randint(nt, (times, n), out=pins[1])                  # straight into pinned memory
        _mmd.apply_row_maps(pins[0], pins[1], ns, nt)
        selection_csr_host(pins[0], ns, 0, 2 * n, out=(pins[2], pins[3]))
        selection_csr_host(pins[1], nt, n, 2 * n, out=(pins[4], pins[5]))

    def _fill_one(self, key):
        e = self._samples[key]
        st = e["group"] if e.get("group") is not None else e        # whose turn / events: the pinned allocation's owner
        k = st["turn"]
        st["turn"] = 1 - k
        if st["done"][k] is not None:
            st["done"][k].synchronize()                             # the copy that last read this block
        self._draw_one(key, k)
        _ship(e["dev"], e["pin"][k])
        ev = torch.cuda.Event()
        ev.record()
        st["done"][k] = ev

    def _provider_dp(self, ns, nt, times, per):
        """Data-parallel branch of MMD(): local row samples + the selection CSRs of their scatter."""
        key = (ns, nt, times, per)
        if key not in self._dp_idx:
            self._dp_idx[key] = _DPSamples(self.src.x.device, ns, nt, times, per)
            self._dp_idx[key].fill()
        return self._dp_idx[key].views()

    def _provider_rand(self, shape):
        """k-th ``host_rand`` call of a step -> slot k (created, and filled, at its first call)."""
        k = self._rand_cursor
        self._rand_cursor += 1
        if k == len(self._rand_slots):
            slot = _RandSlot(shape, self.src.x.device)
            self._rand_slots.append(slot)
            self._order.append(slot.fill)
            slot.fill()
        slot = self._rand_slots[k]
        if slot.shape != shape:
            raise RuntimeError("host_rand call sequence changed between steps: the step cannot be replayed")
        return slot.dev

    def _consolidate_rand(self):
        """Every ``host_rand`` slot of the step as a view of ONE device block and one pair of pinned blocks (after the first
        warm-up run has met them all): a refill is then one host-to-device copy instead of one per call site.  AdaGCN
        draws interpolation weights at ten call sites per step; their ten copies queued in front of every replay cost
        ~320 us of the device's time per epoch (profiles/r6_experiments.txt, 8)."""
        slots = self._rand_slots
        if len(slots) < 2 or getattr(self, "_rand_arena", None) is not None:
            return
        offs, total = [], 0
        for sl in slots:
            offs.append(total)
            total += (sl.dev.numel() + 3) // 4 * 4
        dev = slots[0].dev.device
        arena = dict(dev=torch.zeros(total, dtype=torch.float32, device=dev),
                     pin=[torch.zeros(total, dtype=torch.float32).pin_memory() for _ in range(2)],
                     done=[None, None], turn=0, half=0, dirty=False)
        for sl, o in zip(slots, offs):
            n = sl.dev.numel()
            sl.dev = arena["dev"][o:o + n].view(sl.shape)
            sl.pin = [p[o:o + n].view(sl.shape) for p in arena["pin"]]
            sl.arena = arena
        self._rand_arena = arena

    def _refill(self):
        """The refills of ONE step: sub-step 0's sample blocks, every host_rand slot, the data-parallel blocks."""
        later = {id(f) for u, fs in self._fills.items() if u > 0 for f in fs}
        arena = getattr(self, "_rand_arena", None)
        if arena is not None:
            k = arena["half"] = arena["turn"]
            arena["turn"] = 1 - k
            if arena["done"][k] is not None:
                arena["done"][k].synchronize()
            arena["dirty"] = False
        for fill in self._order:
            if id(fill) not in later:
                fill()
        if arena is not None and arena["dirty"]:
            _ship(arena["dev"], arena["pin"][arena["half"]])
            ev = torch.cuda.Event()
            ev.record()
            arena["done"][arena["half"]] = ev
        for e in self._dp_idx.values():
            e.fill()

    def _refill_multi(self):
        """The refills of `unroll` consecutive steps, in step order (the CPU generator's order in eager mode), shipped as
        ONE copy per MMD call site."""
        if not self._groups:
            for u in range(self.unroll):
                for fill in self._fills.get(u, ()):
                    fill()
            return
        half = {}
        for gk, g in self._groups.items():
            k = half[gk] = g["turn"]
            g["turn"] = 1 - k
            if g["done"][k] is not None:
                g["done"][k].synchronize()
        for u in range(self.unroll):
            for key in self._fill_keys.get(u, ()):
                self._draw_one(key, half[key[:4]])
        for gk, g in self._groups.items():
            _ship(g["dev"], g["pin"][half[gk]])
            ev = torch.cuda.Event()
            ev.record()
            g["done"][half[gk]] = ev

    def _run(self, with_stats=False):
        from .ops import dropout_state
        self._rand_cursor = 0
        # device step counter of the dropout kernels (bumped by every replay too) and, in the same launch when the
        # optimiser allows it, Adam's step counters: their increment then no longer sits in front of the update
        bump = getattr(self.optimizer, "bump_steps", None) if BUMP_AT_START and not self.extra_optimizers else None
        if bump is not None and bump(dropout_state.counter(self.src.x.device)):
            dropout_state.site = 0
        else:
            dropout_state.next_step(self.src.x.device)
        global defer_total
        defer_total = type(self) is GraphedStep and not self.dp
        try:
            loss, logits = self.step_fn(self.src, self.tgt)
        finally:
            defer_total = False
        terms = loss if isinstance(loss, LossTerms) else None
        if terms is not None:
            if getattr(self, "_one", None) is None:
                self._one = torch.ones((), dtype=torch.float32, device=self.src.x.device)
            if not with_stats:                                    # warm-up runs: the total on the main stream
                loss = terms[0].detach()
                for extra in terms[1:]:
                    loss = loss + extra.detach()
        if with_stats:
            # per-epoch numbers of the reference's loop (loss, source micro-F1 = accuracy): two doubles
            # produced inside the graph, on a side branch that runs beside the backward pass
            main = torch.cuda.current_stream()
            side = main if self.inline_stats else self._stat_stream
            if side is not main:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                from .ops import ce_stats_for
                if terms is not None:                             # the reported total: same fp32 sum as `a + b` in eager mode
                    loss = terms[0].detach()
                    for extra in terms[1:]:
                        loss = loss + extra.detach()
                by_product = ce_stats_for(logits, self.src.y)     # the loss kernel counted the correct rows already
                if by_product is not None:
                    by_product[0:1].copy_(loss.detach().reshape(1))   # slot 0: the TOTAL loss of the step
                    self.stats = by_product
                else:
                    correct = (logits.detach().argmax(dim=1) == self.src.y).sum()
                    self.stats = torch.stack([loss.detach().double(), correct.double()])
            if side is not main:
                for t in (loss, logits):
                    t.record_stream(side)
        self.optimizer.zero_grad(set_to_none=True)
        if terms is not None:
            # every term that still has a history is seeded with 1; a term whose backward the trainer has issued already
            # (A2GNN: the cross-entropy chain, on its own stream behind the loss kernel) arrives detached, with the
            # tensor its chain stopped at and that tensor's gradient as a further root, and the gradients of the
            # parameters only that chain reaches
            roots = [t for t in terms if t.requires_grad]
            seeds = [self._one.reshape(t.shape) for t in roots]
            for t, g in getattr(terms, "extra_roots", ()):
                roots.append(t)
                seeds.append(g)
            for p, g in getattr(terms, "preset_grads", ()):
                p.grad = g
            torch.autograd.backward(roots, seeds)
        else:
            loss.backward()
        if self.dp:
            from .distributed import allreduce_grads
            allreduce_grads(p for g in self.optimizer.param_groups for p in g["params"])
        self.optimizer.step()
        if with_stats and side is not main:
            main.wait_stream(side)
            self.stats.record_stream(main)
        return loss, logits

    # -- public ------------------------------------------------------------------------
    def capture(self):
        """Warm-up steps are rolled back afterwards (parameters, Adam moments / step counters,
        the CPU generator), so a seeded fit() takes exactly the steps eager mode would."""
        global host_rand_provider
        prev, prev_dp, prev_rand = _mmd.sample_provider, _mmd.dp_index_provider, host_rand_provider
        _mmd.sample_provider = self._provider
        _mmd.dp_index_provider = self._provider_dp if self.dp else None
        host_rand_provider = self._provider_rand
        optimizers = [self.optimizer] + self.extra_optimizers
        params = [p for o in optimizers for g in o.param_groups for p in g["params"]]
        saved = [p.detach().clone() for p in params]
        cpu_rng = torch.get_rng_state()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(self.warmup):       # allocator warm-up + first sample buffers
                    self._refill()
                    self._run()
                    if i == 0:
                        self._consolidate_rand()
                if self.unroll > 1 and (self._rand_slots or self.extra_optimizers or not self._fills):
                    self.unroll = 1                # host_rand call sites / critics: one step per capture
                for u in range(1, self.unroll):    # the further sub-steps' sample blocks are created (and filled) eagerly
                    self._sub = u
                    self._run()
                self._sub = 0
            torch.cuda.current_stream().wait_stream(side)
            self._refill()
            # thread_local: API calls of other threads (RCCL's watchdog polls events) must not
            # invalidate the capture
            self._stat_stream = torch.cuda.Stream()
            self.graph_multi = None
            if self.unroll > 1:
                self.graph_multi = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_multi, capture_error_mode="thread_local"):
                    per_step = []
                    for u in range(self.unroll):
                        self._sub = u
                        self._run(with_stats=True)
                        per_step.append(self.stats)
                    self.stats_multi = torch.stack(per_step)   # [unroll, 2]: ONE small D2H per replay
                self._sub = 0
                self._multi_pins = [torch.zeros(self.unroll, 2, dtype=torch.float64).pin_memory() for _ in range(2)]
                self._multi_events, self._multi_turn = [None, None], 0
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                loss, logits = self._run(with_stats=True)      # an epoch then needs ONE small D2H
            self.loss, self.logits = loss.detach(), logits.detach()
            self._preroll()
            self._stat_pins = [torch.zeros(2, dtype=torch.float64).pin_memory() for _ in range(2)]
            self._stat_events = [None, None]
            self._stat_turn = 0
            with torch.no_grad():
                for p, v in zip(params, saved):
                    p.copy_(v)
                for o in optimizers:                           # fresh optimisers: everything back to zero
                    for st in o.state.values():
                        for v in st.values():
                            if torch.is_tensor(v):
                                v.zero_()
            torch.set_rng_state(cpu_rng)
        finally:
            _mmd.sample_provider, _mmd.dp_index_provider, host_rand_provider = prev, prev_dp, prev_rand
        return self

    def _preroll(self):
        """Between capture and roll-back: hand the fresh executable graphs to the device ahead of their first timed
        launch (``hipGraphUpload``) and / or replay them a few times (``PYGDA_AMD_GRAPH_PREROLL=R``) -- everything a replay
        changes (parameters, optimiser state) is rolled back right after, the sample blocks are not refilled.  Why:
        every long run met ONE replay of 5-8 ms early in a graph's life in which the host's launch call blocks
        (profiles/r4_replay_jitter.txt); with four steps per capture it fell inside a 20-step timed region."""
        import os
        graphs = [g for g in (getattr(self, "graph_multi", None), getattr(self, "graph", None),
                               *getattr(self, "graphs", ())) if g is not None]
        if os.environ.get("PYGDA_AMD_GRAPH_UPLOAD", "1") == "1":
            try:
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                hip.hipGraphUpload.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
                hip.hipGraphUpload.restype = ctypes.c_int
                stream = torch.cuda.current_stream().cuda_stream
                self.upload_status = [int(hip.hipGraphUpload(ctypes.c_void_p(g.raw_cuda_graph_exec()),
                                                             ctypes.c_void_p(stream))) for g in graphs]
            except Exception as exc:                 # noqa: BLE001 -- an optimisation hint: never a reason to fail
                self.upload_status = f"{type(exc).__name__}: {exc}"
        rolls = int(os.environ.get("PYGDA_AMD_GRAPH_PREROLL", "0"))
        for g in graphs:
            for _ in range(rolls):
                g.replay()
        if rolls:
            torch.cuda.synchronize()

    def _replay(self):
        self.graph.replay()

    def __call__(self):
        self._refill()
        self._replay()
        return self.loss, self.logits

    # -- pipelined epochs: launch step e+1 before reading the numbers of step e --------------
    def launch(self):
        """Refill + replay + asynchronous read-back of (loss, correct); returns a ticket."""
        self._refill()
        self._replay()
        k = self._stat_turn
        self._stat_turn = 1 - k
        self._stat_pins[k].copy_(self.stats, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._stat_events[k] = ev
        return k

    def result(self, ticket):
        """(loss, source accuracy) of the step behind ``ticket`` (at most one newer launch may exist)."""
        if isinstance(ticket, tuple):
            return self.result_multi(ticket)[0]
        self._stat_events[ticket].synchronize()
        loss, correct = self._stat_pins[ticket].tolist()
        n = self.src.y.numel()
        return loss, (correct / n if n else 0.0)

    def launch_multi(self):
        """`unroll` steps in one replay; returns a ticket for result_multi()."""
        self._refill_multi()
        self.graph_multi.replay()
        k = self._multi_turn
        self._multi_turn = 1 - k
        self._multi_pins[k].copy_(self.stats_multi, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._multi_events[k] = ev
        return ("multi", k)

    def result_multi(self, ticket):
        """[(loss, source accuracy)] * unroll of the replay behind ``ticket``, in step order."""
        k = ticket[1]
        self._multi_events[k].synchronize()
        n = self.src.y.numel()
        return [(loss, (correct / n if n else 0.0)) for loss, correct in self._multi_pins[k].tolist()]


class GraphedStepSplit(GraphedStep):
    """The step as THREE hipGraphs: source forward (stream A) and target forward (stream B) replayed at the same
    time, then -- on the main stream -- domain loss, epoch statistics, backward through both tapes and the
    optimiser step.

    Why: on this runtime a forked capture replays with 5-6 us of spacing per kernel and runs branches forked at
    its root one after the other (tools/graph_fork_*.py, profiles/HISTORY.md 4.7) -- the one-graph A2GNN step spends the first
    110 us on the source branch alone.  Single-branch graphs replay gap-free through pre-built AQL packets, and two
    graph launches on two streams do overlap.  Autograd's tape spans the captures (the technique of
    torch.cuda.make_graphed_callables and of GraphedStepDP): the forward graphs keep their outputs and saved
    tensors alive in their own pools, the third capture walks the tape backwards (autograd runs each node on its
    forward stream, so the backward of the two branches forks inside the third graph, mid-chain, where the
    runtime does overlap).  The dropout step counter is bumped eagerly in front of the three launches.

    ``parts = (src_part(src), tgt_part(tgt), rest_part(src_out, tgt_out, alpha))`` come from the trainer."""

    def __init__(self, parts, scalar_alpha, step_fn, optimizer, src, tgt, warmup=3):
        super().__init__(step_fn, optimizer, src, tgt, warmup=warmup)
        self.parts, self.alpha = parts, scalar_alpha

    def _run_split(self, capture):
        """One step through the three parts; ``capture`` = the three CUDAGraph objects or None (eager)."""
        from .ops import dropout_state
        src_part, tgt_part, rest_part = self.parts
        main = torch.cuda.current_stream()
        mode = dict(capture_error_mode="thread_local")
        dropout_state.site = 0                       # call sites number through the three captures
        if capture is None:
            a = src_part(self.src)
            b = tgt_part(self.tgt)
            return self._tail(rest_part, a, b, True)
        g1, g2, g3 = capture
        with torch.cuda.graph(g1, stream=self._sa, **mode):
            a = src_part(self.src)
        with torch.cuda.graph(g2, stream=self._sb, **mode):
            b = tgt_part(self.tgt)
        with torch.cuda.graph(g3, **mode):
            out = self._tail(rest_part, a, b, True)
        self._keep = (a, b)                          # the forward graphs' outputs stay alive with the graphs
        return out

    def _tail(self, rest_part, a, b, with_stats):
        loss, logits = rest_part(a, b, self.alpha)       # a plain loss tensor: `defer_total` is set for GraphedStep only
        main = torch.cuda.current_stream()
        if with_stats:
            side = self._stat_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                from .ops import ce_stats_for
                by_product = ce_stats_for(logits, self.src.y)     # the loss kernel counted the correct rows already
                if by_product is not None:
                    by_product[0:1].copy_(loss.detach().reshape(1))    # slot 0: the TOTAL loss of the step
                    self.stats = by_product
                else:
                    correct = (logits.detach().argmax(dim=1) == self.src.y).sum()
                    self.stats = torch.stack([loss.detach().double(), correct.double()])
            if side is not main:
                for t in (loss, logits):
                    t.record_stream(side)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        if with_stats and side is not main:
            main.wait_stream(side)
            self.stats.record_stream(main)
        return loss, logits

    def capture(self):
        from .ops import dropout_state
        prev = _mmd.sample_provider
        _mmd.sample_provider = self._provider
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        saved = [p.detach().clone() for p in params]
        cpu_rng = torch.get_rng_state()
        dev = self.src.x.device
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):           # allocator warm-up, first sample buffers, K-step plans
                    self._refill()
                    self._run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._refill()
            self._sa, self._sb, self._stat_stream = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
            self.graphs = tuple(torch.cuda.CUDAGraph() for _ in range(3))
            dropout_state.next_step(dev)               # eager, as in front of every replay
            loss, logits = self._run_split(self.graphs)
            self.loss, self.logits = loss.detach(), logits.detach()
            self._preroll()
            self._stat_pins = [torch.zeros(2, dtype=torch.float64).pin_memory() for _ in range(2)]
            self._stat_events = [None, None]
            self._stat_turn = 0
            with torch.no_grad():
                for p, v in zip(params, saved):
                    p.copy_(v)
                for st in self.optimizer.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
            torch.set_rng_state(cpu_rng)
        finally:
            _mmd.sample_provider = prev
        return self

    def _replay(self):
        from .ops import dropout_state
        main = torch.cuda.current_stream()
        dropout_state.next_step(self.src.x.device)     # fresh dropout masks: the counter all three graphs read
        g1, g2, g3 = self.graphs
        self._sa.wait_stream(main)
        self._sb.wait_stream(main)
        with torch.cuda.stream(self._sa):
            g1.replay()
        with torch.cuda.stream(self._sb):
            g2.replay()
        main.wait_stream(self._sa)
        main.wait_stream(self._sb)
        g3.replay()


class GraphedStepDP:
    """Data-parallel variant: RCCL collectives cannot be stream-captured on this stack, so the
    step is cut at its two exchange points into three hipGraphs (G2 and G3 below are one capture) with the
    collectives launched eagerly between them (same stream, so ordering is automatic):

        G1  encoders, source CE, local MMD row samples            (forward, autograd tape kept)
        --  all_gather(source rows), all_gather(target rows)      (RCCL)
        G2  global-batch MMD forward + its gradient w.r.t. the rows; slice own rows, x W
        G3  backward of G1 from (CE, row gradients) -> flat gradient buffer
        --  all_reduce(flat gradients)                            (RCCL)
        G4  average, Adam

    ``part1(src, tgt, idx_s, idx_t) -> (loss_ce, source_logits, rows_s, rows_t)`` and
    ``part2(rows_s_all, rows_t_all) -> weighted mmd`` come from the trainer.  The graphs share
    one memory pool, the technique of torch.cuda.make_graphed_callables (forward and backward of
    a section as separate graphs over static tensors)."""

    def __init__(self, part1, part2, optimizer, src, tgt, times=5, sampling_num=1000):
        import torch.distributed as dist
        self.part1, self.part2, self.optimizer, self.src, self.tgt = part1, part2, optimizer, src, tgt
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.times, self.per = times, -(-sampling_num // self.world)
        self.samples = _DPSamples(src.x.device, src.x.size(0), tgt.x.size(0), times, self.per)
        self.loss = self.logits = None

    def _refill(self):                       # the eager data-parallel MMD()'s draws, same order
        self.samples.fill()

    def capture(self, eager_step, warmup=2):
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        saved = [p.detach().clone() for p in params]
        cpu_rng = torch.get_rng_state()
        for _ in range(warmup):              # RCCL + allocator + optimiser-state warm-up, rolled back below
            eager_step()
        torch.cuda.synchronize()
        self._refill()
        dev = self.src.x.device
        W, rank = self.world, self.rank
        g1, g2, g4 = (torch.cuda.CUDAGraph() for _ in range(3))
        mode = dict(capture_error_mode="thread_local")    # RCCL's watchdog thread polls events meanwhile
        one = torch.ones((), dtype=torch.float32, device=dev)
        with torch.cuda.graph(g1, **mode):
            from .ops import dropout_state
            dropout_state.next_step(dev)
            loss_ce, logits, rows_s, rows_t = self.part1(self.src, self.tgt, *self.samples.views())
            rows_st = torch.stack([rows_s.detach(), rows_t.detach()])      # ONE all-gather for both domains
        pool = g1.pool()
        self.gath = torch.zeros((W,) + tuple(rows_st.shape), dtype=torch.float32, device=dev, requires_grad=True)
        d = rows_s.size(-1)
        with torch.cuda.graph(g2, pool=pool, **mode):
            S = self.gath[:, 0].permute(1, 0, 2, 3).reshape(self.times, W * self.per, d)
            T = self.gath[:, 1].permute(1, 0, 2, 3).reshape(self.times, W * self.per, d)
            dom = self.part2(S, T)
            (gG,) = torch.autograd.grad(dom, [self.gath])
            g_rows_s, g_rows_t = gG[rank, 0] * float(W), gG[rank, 1] * float(W)
            total = loss_ce.detach() + dom.detach()
            # (G3 continues in the same capture: nothing is exchanged between the two, and one graph launch less per
            # step is ~30 us at these kernel sizes)
            grads = torch.autograd.grad([loss_ce, rows_s, rows_t], params, [one, g_rows_s, g_rows_t],
                                        allow_unused=True)
            # every gradient in its parameter's MEMORY order (a weight stored gather-major, sparse_features.py, has
            # transposed strides: flattening it logically would transpose it here and back again in front of Adam)
            flat = torch.cat([_mem_flat(g if g is not None else torch.zeros_like(p), p) for g, p in zip(grads, params)])
        off = 0
        for p in params:                     # the optimiser reads static views of the reduced buffer
            p.grad = _mem_view(flat[off:off + p.numel()], p)
            off += p.numel()
        with torch.cuda.graph(g4, pool=pool, **mode):
            flat.div_(float(W))
            self.optimizer.step()
            correct = (logits.detach().argmax(dim=1) == self.src.y).sum()
            self.stats = torch.stack([total.double(), correct.double()])   # this replica's epoch numbers
        self._graphs, self._rows_st, self._flat = (g1, g2, g4), rows_st, flat
        # every tensor a captured kernel reads must outlive the graphs: `one` in particular lives in
        # the ordinary pool, and once freed its block would be recycled by eager allocations
        self._keep = (one, loss_ce, logits, rows_s, rows_t, gG, g_rows_s, g_rows_t, dom, total, grads, correct)
        self.loss, self.logits = total, logits.detach()
        self._stat_pins = [torch.zeros(2, dtype=torch.float64).pin_memory() for _ in range(2)]
        self._stat_events, self._stat_turn = [None, None], 0
        with torch.no_grad():                # roll the warm-up back: a seeded fit() takes the eager steps
            for p, v in zip(params, saved):
                p.copy_(v)
            for st in self.optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        torch.set_rng_state(cpu_rng)
        self.graph = g1
        return self

    def __call__(self):
        import torch.distributed as dist
        g1, g2, g4 = self._graphs
        self._refill()
        g1.replay()
        with torch.no_grad():
            # flat views: RCCL takes any output of W x the input's size; gloo (CPU tests, bench --share-gpus) chunks
            # the output along dim 0 and wants each chunk in the input's shape
            dist.all_gather_into_tensor(self.gath.view(-1), self._rows_st.view(-1))
        g2.replay()
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
        g4.replay()
        return self.loss, self.logits

    # -- pipelined epochs (see GraphedStep.launch / result) ----------------------------------
    def launch(self):
        self()
        k = self._stat_turn
        self._stat_turn = 1 - k
        self._stat_pins[k].copy_(self.stats, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._stat_events[k] = ev
        return k

    def result(self, ticket):
        self._stat_events[ticket].synchronize()
        loss, correct = self._stat_pins[ticket].tolist()
        n = self.src.y.numel()
        return loss, (correct / n if n else 0.0)
