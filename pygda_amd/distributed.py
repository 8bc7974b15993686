"""Data-parallel plumbing over ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
the GPU box, ``gloo`` in the CPU tests).  One process per GPU; graph and features are
replicated; ranks own disjoint seed mini-batches (``NeighborLoader(rank, world_size)``).
Two exchange steps per training step, both small and latency-bound:

* one flat all-reduce of every gradient (averaged over ranks);
* an all-gather of the MMD sample rows, so that the pairwise loss is taken over the GLOBAL
  batch (the cross-domain statistic cannot be formed from per-rank losses).
"""
import ctypes
import os

import torch
import torch.distributed as dist


class _DirectComm:
    """RCCL communicator owned by libgda_hip.so (include/gda_hip.h: gda_comm_*): the two exchange
    steps become plain enqueues on the current stream -- no ProcessGroup work objects, no watchdog
    events -- which is what lets a whole data-parallel step be captured into ONE hipGraph.
    Default for ``nccl`` groups (:func:`direct`); the rendezvous (shipping the 128-byte id) still rides on
    the torch.distributed group the launcher set up."""

    def __init__(self):
        from . import _lib
        self.L = L = _lib.lib()
        self._lib = _lib
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _lib.check(L.gda_rccl_load(path.encode() if os.path.isfile(path) else None), "gda_rccl_load")
        dev = torch.device("cuda", torch.cuda.current_device())
        uid = torch.zeros(128, dtype=torch.uint8)
        if dist.get_rank() == 0:
            buf = (ctypes.c_char * 128)()
            _lib.check(L.gda_comm_unique_id(buf, 128), "gda_comm_unique_id")
            uid = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()
        uid = uid.to(dev)
        dist.broadcast(uid, src=0)
        raw = bytes(uid.cpu().tolist())
        handle = ctypes.c_void_p()
        _lib.check(L.gda_comm_init_rank(raw, 128, dist.get_world_size(), dist.get_rank(), ctypes.byref(handle)),
                   "gda_comm_init_rank")
        self.handle = handle

    def all_reduce_(self, flat):
        self._lib.check(self.L.gda_allreduce_f32(self._lib.ptr(flat), flat.numel(), self.handle,
                                                 self._lib.stream()), "gda_allreduce_f32")

    def all_gather(self, out, x):
        self._lib.check(self.L.gda_allgather_f32(self._lib.ptr(x), self._lib.ptr(out), x.numel(), self.handle,
                                                 self._lib.stream()), "gda_allgather_f32")


_direct = None
_direct_failed = None       # why the library-owned communicator was given up for this process (str), or None


def _self_test(comm, timeout_s=20.0):
    """One all-reduce and one all-gather on the fresh communicator, checked against what they must return; polled with
    a deadline so that a communicator that never completes costs the process `timeout_s`, not the run."""
    import time
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = torch.full((257,), float(rank + 1), dtype=torch.float32, device=dev)
    g = torch.empty(world, 3, dtype=torch.float32, device=dev)
    comm.all_reduce_(x)
    comm.all_gather(g, torch.full((3,), float(rank), dtype=torch.float32, device=dev))
    ev = torch.cuda.Event()
    ev.record()
    t0 = time.time()
    while not ev.query():
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"self-test collectives did not complete within {timeout_s:.0f} s")
        time.sleep(0.001)
    want = world * (world + 1) / 2.0
    if not bool((x == want).all()):
        raise RuntimeError(f"all-reduce self-test: got {float(x[0])}, expected {want}")
    if not bool((g == torch.arange(world, dtype=torch.float32, device=dev).view(-1, 1)).all()):
        raise RuntimeError("all-gather self-test: rank order of the gathered blocks is wrong")


def direct():
    """The library-owned RCCL communicator (csrc/gda_comm.cpp): the DEFAULT for the exchange steps of an ``nccl``
    group since round 4 -- plain enqueues on the caller's stream, no ProcessGroup work objects, no per-collective
    host bookkeeping.  It is built on first use (unique id shipped over the torch.distributed group) and must pass a
    self-test (all-reduce + all-gather against known answers, with a deadline); ANY failure -- librccl symbols not
    found, communicator init, wrong answer, timeout -- is reported once and every collective of this process then goes
    through torch.distributed's ProcessGroup (same RCCL underneath), which stays the fallback.
    ``PYGDA_AMD_RCCL_DIRECT=0`` switches it off.  gloo groups (CPU tests, ranks sharing a GPU) never use it."""
    global _direct, _direct_failed
    if (os.environ.get("PYGDA_AMD_RCCL_DIRECT", "1") == "0" or _direct_failed is not None or not active()
            or dist.get_backend() != "nccl"):
        return None
    if _direct is None:
        try:
            comm = _DirectComm()
            _self_test(comm)
            _direct = comm
        except Exception as exc:                     # noqa: BLE001 -- anything: the ProcessGroup path is complete
            _direct_failed = f"{type(exc).__name__}: {exc}"
            import warnings
            warnings.warn("library-owned RCCL communicator unavailable (" + _direct_failed +
                          "); collectives go through torch.distributed")
            return None
        # every rank must agree (a rank on the ProcessGroup path and a rank on the communicator would deadlock):
        # the agreement itself rides on the ProcessGroup
    return _direct


def direct_agreed():
    """Collective: build the communicator on every rank and keep it only if EVERY rank succeeded (one rank falling back
    alone would leave its peers waiting in a collective it never joins).  Call once, at start-up, from all ranks."""
    global _direct, _direct_failed
    if not active() or dist.get_backend() != "nccl":
        return None
    ok = torch.tensor([1.0 if direct() is not None else 0.0], device=torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok) < 1.0 and _direct is not None:
        _direct_failed = "another rank could not build its communicator"
        _direct = None
    return _direct


def capture_collectives():
    """Whether a data-parallel step may be captured WITH its collectives into one hipGraph (opt-in,
    ``PYGDA_AMD_RCCL_CAPTURE=1``: validated on a 1-rank group only -- no multi-GPU node was available to any round --
    so N > 1 runs keep the collectives eager between captured segments unless asked otherwise)."""
    return os.environ.get("PYGDA_AMD_RCCL_CAPTURE") == "1" and direct() is not None


def shutdown_direct():
    """Destroy the library-owned communicator (tests; normal runs keep it for the process lifetime)."""
    global _direct
    if _direct is not None:
        torch.cuda.synchronize()
        _direct._lib.check(_direct.L.gda_comm_destroy(_direct.handle), "gda_comm_destroy")
        _direct = None


def info():
    if dist.is_available() and dist.is_initialized():
        return dict(rank=dist.get_rank(), world_size=dist.get_world_size())
    return dict(rank=0, world_size=1)


def active():
    """True when the data-parallel exchange steps must run.  ``PYGDA_AMD_FORCE_DP=1`` switches them
    on for a 1-rank group too (single-GPU test of the multi-GPU code path)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PYGDA_AMD_FORCE_DP") == "1"


def broadcast_parameters(module, src=0):
    """Every replica starts from rank ``src``'s parameters and buffers (what DistributedDataParallel does
    at construction): ranks may have seeded their generators differently (dropout, MMD samples)."""
    if not active():
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers() if b.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    if flat.is_cuda and dist.get_backend() != "nccl":      # gloo group over device tensors: via the host
        h = flat.cpu()
        dist.broadcast(h, src=src)
        flat.copy_(h)
    else:
        dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def _all_reduce_sum(t):
    """In-place sum over ranks.  ``gloo`` groups (CPU tests; two processes sharing one GPU in the
    GPU equality test) stage device tensors through the host instead of relying on gloo's optional
    device support."""
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class _GlobalMean(torch.autograd.Function):
    """Mean over the rows of ALL ranks from each rank's local mean and row count.

    Ranks own sub-graphs of different sizes, so the average of per-rank means is not the mean over
    the global batch.  Forward: ``sum_r n_r * mean_r / sum_r n_r`` (one 2-float
    all-reduce).  Backward: the gradient of that global mean w.r.t. this rank's local mean is
    ``n_r / N``; it is multiplied by W because :func:`allreduce_grads` later averages over ranks --
    the averaged gradient then equals the 1-rank gradient on the concatenated batch."""

    @staticmethod
    def forward(ctx, mean_local, n_local):
        if int(n_local) == 0:      # a rank without rows: its "mean" is 0/0; it contributes nothing and receives nothing
            mean_local = torch.zeros_like(mean_local)
        # (the row count enters as a fill and as scalar kernel arguments: a tensor made from a host number is a
        # pageable H2D copy, i.e. a stream synchronisation per call, and this runs once per data-parallel step)
        buf = torch.stack([mean_local.detach().to(torch.float32).reshape(()) * float(n_local),
                           torch.full((), float(n_local), dtype=torch.float32, device=mean_local.device)])
        _all_reduce_sum(buf)
        ctx.scale = (float(n_local) * dist.get_world_size()) / buf[1]
        ctx.empty = int(n_local) == 0
        return (buf[0] / buf[1]).to(mean_local.dtype).reshape(mean_local.shape)

    @staticmethod
    def backward(ctx, g):
        if ctx.empty:              # never NaN * 0: the upstream mean of zero rows may carry a NaN gradient path
            return torch.zeros_like(g), None
        return g * ctx.scale.to(g.dtype), None


def global_mean(mean_local, n_local):
    """``mean_local``: a 0-dim loss that is a mean over ``n_local`` rows of this rank's batch ->
    the mean over every rank's rows, differentiable (identity in single-process runs)."""
    if not active():
        return mean_local
    return _GlobalMean.apply(mean_local, int(n_local))


def allreduce_grads(params):
    """ONE collective per step: flatten, all-reduce(sum), divide by the world size."""
    if not active():
        return
    params = list(params)
    for p in params:                      # a rank whose batch never touched p still joins
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    comm = direct() if flat.dtype == torch.float32 and flat.is_cuda else None
    if comm is not None:
        comm.all_reduce_(flat)
    else:
        _all_reduce_sum(flat)
    flat.div_(dist.get_world_size())
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


class _AllGatherRows(torch.autograd.Function):
    """``[k, ...]`` per rank -> ``[W, k, ...]`` on every rank.  Every rank then evaluates the
    SAME global loss on the gathered rows, so the gradient of that single objective w.r.t. the
    rows a rank owns is just its slice of the incoming gradient -- no collective in backward.
    The slice is multiplied by W because :func:`allreduce_grads` later averages over ranks."""

    @staticmethod
    def forward(ctx, x):
        w = dist.get_world_size()
        out = torch.empty((w,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        comm = direct() if x.dtype == torch.float32 and x.is_cuda else None
        if comm is not None:
            comm.all_gather(out, x.contiguous())
        elif dist.get_backend() == "nccl":     # single-buffer form: no staging copies
            dist.all_gather_into_tensor(out, x.contiguous())
        elif x.is_cuda:                        # gloo group over device tensors: via the host
            parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(w)]
            dist.all_gather(parts, x.detach().cpu().contiguous())
            out.copy_(torch.stack(parts))
        else:
            dist.all_gather(list(out.unbind(0)), x.contiguous())
        ctx.rank, ctx.world = dist.get_rank(), w
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank] * float(ctx.world)


def all_gather_rows(x):
    return _AllGatherRows.apply(x) if active() else x.unsqueeze(0)
