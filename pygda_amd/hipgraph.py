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
import torch

from .utils import mmd as _mmd


class GraphedStep:
    """Captures ``loss, logits = step_fn(src, tgt)``, ``backward`` and ``optimizer.step()``."""

    def __init__(self, step_fn, optimizer, src, tgt, warmup=3):
        self.step_fn, self.optimizer, self.src, self.tgt = step_fn, optimizer, src, tgt
        self._samples = {}                # (ns, nt, times, n) -> (dev_s, dev_t, pin_s, pin_t)
        self._order = []
        self.graph = None
        self.warmup = warmup
        self.loss = self.logits = None

    # -- MMD sample plumbing ---------------------------------------------------------
    def _provider(self, ns, nt, times, n):
        """Static device buffers: the row samples and the selection CSRs of their scatter."""
        key = (ns, nt, times, n)
        if key not in self._samples:
            dev = self.src.x.device
            shapes = [((times, n), torch.int64), ((times, n), torch.int64), ((ns + 1,), torch.int32),
                      ((times * n,), torch.int32), ((nt + 1,), torch.int32), ((times * n,), torch.int32)]
            devb = [torch.zeros(sh, dtype=dt, device=dev) for sh, dt in shapes]
            pins = [torch.zeros(sh, dtype=dt).pin_memory() for sh, dt in shapes]
            ones = torch.ones(times * n, dtype=torch.float32, device=dev)
            self._samples[key] = (devb, pins, ones)
            self._order.append(key)
            self._fill_one(key)
        devb, _, ones = self._samples[key]
        return devb[0], devb[1], (devb[2], devb[3], devb[4], devb[5], ones)

    def _fill_one(self, key):
        from .ops import selection_csr_host
        ns, nt, times, n = key
        devb, pins, _ = self._samples[key]
        torch.randint(ns, (times, n), out=pins[0])                  # eager MMD()'s draws, same order,
        torch.randint(nt, (times, n), out=pins[1])                  # straight into pinned memory
        selection_csr_host(pins[0], ns, 0, 2 * n, out=(pins[2], pins[3]))
        selection_csr_host(pins[1], nt, n, 2 * n, out=(pins[4], pins[5]))
        for d, p in zip(devb, pins):
            d.copy_(p, non_blocking=True)

    def _refill(self):
        for key in self._order:
            self._fill_one(key)

    def _run(self):
        loss, logits = self.step_fn(self.src, self.tgt)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss, logits

    # -- public ------------------------------------------------------------------------
    def capture(self):
        """Warm-up steps are rolled back afterwards (parameters, Adam moments / step counters,
        the CPU generator), so a seeded fit() takes exactly the steps eager mode would."""
        prev = _mmd.sample_provider
        _mmd.sample_provider = self._provider
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        saved = [p.detach().clone() for p in params]
        cpu_rng = torch.get_rng_state()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):       # allocator warm-up + first sample buffers
                    self._refill()
                    self._run()
            torch.cuda.current_stream().wait_stream(side)
            self._refill()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                loss, logits = self._run()
            self.loss, self.logits = loss.detach(), logits.detach()
            with torch.no_grad():
                for p, v in zip(params, saved):
                    p.copy_(v)
                for st in self.optimizer.state.values():      # fresh optimiser: everything back to zero
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
            torch.set_rng_state(cpu_rng)
        finally:
            _mmd.sample_provider = prev
        return self

    def __call__(self):
        self._refill()
        self.graph.replay()
        return self.loss, self.logits
