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
    OPT-IN (``PYGDA_AMD_RCCL_DIRECT=1``): it has only ever been exercised on a 1-rank group.  Built in stages by
    :func:`direct_agreed`, every stage followed by an agreement over the torch.distributed group, so that a rank
    that fails alone never leaves its peers in a collective it does not join."""

    def __init__(self):
        from . import _lib
        self.L = _lib.lib()
        self._lib = _lib
        self.handle = None
        self.uid = None

    def load(self):
        """Stage 1 (local, no collective): bind librccl; rank 0 makes the unique id."""
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._lib.check(self.L.gda_rccl_load(path.encode() if os.path.isfile(path) else None), "gda_rccl_load")
        self.uid = torch.zeros(128, dtype=torch.uint8)
        if dist.get_rank() == 0:
            buf = (ctypes.c_char * 128)()
            self._lib.check(self.L.gda_comm_unique_id(buf, 128), "gda_comm_unique_id")
            self.uid = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()

    def init_rank(self):
        """Stage 2 (collective; entered only when EVERY rank passed stage 1): ship the id, join the communicator."""
        dev = torch.device("cuda", torch.cuda.current_device())
        uid = self.uid.to(dev)
        dist.broadcast(uid, src=0)
        raw = bytes(uid.cpu().tolist())
        handle = ctypes.c_void_p()
        self._lib.check(self.L.gda_comm_init_rank(raw, 128, dist.get_world_size(), dist.get_rank(),
                                                  ctypes.byref(handle)), "gda_comm_init_rank")
        self.handle = handle

    def destroy(self):
        if self.handle is not None:
            try:
                self._lib.check(self.L.gda_comm_destroy(self.handle), "gda_comm_destroy")
            finally:
                self.handle = None

    def all_reduce_(self, flat, stream=None):
        self._lib.check(self.L.gda_allreduce_f32(self._lib.ptr(flat), flat.numel(), self.handle,
                                                 self._lib.stream() if stream is None else stream), "gda_allreduce_f32")

    def all_gather(self, out, x, stream=None):
        self._lib.check(self.L.gda_allgather_f32(self._lib.ptr(x), self._lib.ptr(out), x.numel(), self.handle,
                                                 self._lib.stream() if stream is None else stream), "gda_allgather_f32")


_direct = None
_direct_failed = None       # why the library-owned communicator was given up for this process (str), or None
_direct_tried = False


def _self_test(comm, timeout_s=20.0):
    """One all-reduce and one all-gather on the fresh communicator, checked against what they must return.  They run
    on a stream of their OWN (a self-test that never completes must not leave its collectives queued in front of the
    training stream's work) and are polled with a deadline, so a dead communicator costs `timeout_s`, not the run."""
    import time
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        x = torch.full((257,), float(rank + 1), dtype=torch.float32, device=dev)
        g = torch.empty(world, 3, dtype=torch.float32, device=dev)
        mine = torch.full((3,), float(rank), dtype=torch.float32, device=dev)
        comm.all_reduce_(x, stream=side.cuda_stream)
        comm.all_gather(g, mine, stream=side.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(side)
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


def direct_enabled():
    """The library-owned communicator is OPT-IN (``PYGDA_AMD_RCCL_DIRECT=1``): no round had a
    multi-GPU node, so it has never seen two ranks.  The torch.distributed ProcessGroup (the same RCCL underneath)
    is the default for every collective."""
    return os.environ.get("PYGDA_AMD_RCCL_DIRECT", "0") == "1"


def direct():
    """The communicator :func:`direct_agreed` built and every rank agreed on, or None.  Never builds one itself: a
    rank constructing it alone, in the middle of a step, is exactly the asymmetric rendezvous that hangs a job."""
    if not direct_enabled() or not active() or dist.get_backend() != "nccl":
        return None
    return _direct


def _all_ranks_ok(local_ok):
    flag = torch.tensor([1.0 if local_ok else 0.0], device=torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return float(flag) >= 1.0


def direct_agreed():
    """Collective, called once at start-up by ALL ranks (``fit()``, ``bench.py``): build the library-owned RCCL
    communicator in three stages -- bind librccl + unique id | id broadcast + ``ncclCommInitRank`` | self-test --
    each followed by an all-reduce(MIN) of "this rank succeeded" over the ProcessGroup.  A failure on ANY rank at ANY
    stage makes EVERY rank leave at the same point: the ranks' ProcessGroup collectives always match (the hang
    ADVICE round 4 describes -- one rank skipping the id broadcast -- cannot happen), a half-built communicator is
    destroyed, and every collective of the run stays on the ProcessGroup path.  Returns the communicator or None."""
    global _direct, _direct_failed, _direct_tried
    if not direct_enabled() or not active() or dist.get_backend() != "nccl":
        return None
    if _direct_tried:
        return _direct
    _direct_tried = True
    import warnings
    comm, err = None, None

    def stage(name, fn):
        nonlocal err
        if err is None:
            try:
                fn()
            except Exception as exc:                 # noqa: BLE001 -- anything: the ProcessGroup path is complete
                err = f"{name}: {type(exc).__name__}: {exc}"
        ok = _all_ranks_ok(err is None)
        if not ok and err is None:
            err = f"{name}: another rank failed"
        return ok

    def make():
        nonlocal comm
        comm = _DirectComm()
        comm.load()

    good = stage("load", make) and stage("init_rank", lambda: comm.init_rank()) and \
        stage("self_test", lambda: _self_test(comm))
    if good:
        _direct = comm
    else:
        _direct_failed = err
        if comm is not None:
            try:
                comm.destroy()
            except Exception:                        # noqa: BLE001
                pass
        warnings.warn("library-owned RCCL communicator unavailable (" + str(err) +
                      "); collectives go through torch.distributed")
    return _direct


def capture_collectives():
    """Whether a data-parallel step may be captured WITH its collectives into one hipGraph (opt-in,
    ``PYGDA_AMD_RCCL_CAPTURE=1`` on top of ``PYGDA_AMD_RCCL_DIRECT=1``: validated on a 1-rank group only -- no
    multi-GPU node was available to any round -- so N > 1 runs keep the collectives eager between captured segments
    unless asked otherwise)."""
    return os.environ.get("PYGDA_AMD_RCCL_CAPTURE") == "1" and direct() is not None


def shutdown_direct():
    """Destroy the library-owned communicator (tests; normal runs keep it for the process lifetime)."""
    global _direct, _direct_tried, _direct_failed
    if _direct is not None:
        torch.cuda.synchronize()
        _direct.destroy()
        _direct = None
    _direct_tried, _direct_failed = False, None


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


_coalesced_state = None       # None: unchecked, "on": first-step check passed, "off": it failed (flat buffer from then on)


def _dense_view(g):
    """A contiguous view of a dense gradient's memory (a gather-major weight's gradient has transposed strides)."""
    if g.is_contiguous():
        return g
    if g.dim() == 2 and g.t().is_contiguous():
        return g.t()
    return None


def allreduce_grads(params):
    """ONE collective launch per step: every gradient summed over the ranks IN PLACE, then divided by the world size.

    ``nccl`` groups: the all-reduces of the individual gradients are issued inside one coalescing group (RCCL fuses a
    group into a single launch) and divided by one multi-tensor kernel -- no flattened copy, no copy-back per parameter
    (round 4: ``torch.cat`` + W + one ``copy_`` per parameter, ~2 N launches).  The library-owned
    communicator and gloo groups (CPU tests; ranks sharing a GPU) keep one flat buffer: a single contiguous operand is
    what those paths take."""
    if not active():
        return
    params = list(params)
    for p in params:                      # a rank whose batch never touched p still joins
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    world = dist.get_world_size()
    grads = [p.grad for p in params]
    cuda32 = all(g.is_cuda and g.dtype == torch.float32 for g in grads)
    comm = direct() if cuda32 else None
    views = [_dense_view(g) for g in grads] if cuda32 else None
    manager = getattr(dist, "_coalescing_manager", None)
    global _coalesced_state
    want = os.environ.get("PYGDA_AMD_COALESCED_GRADS", "check")        # "1": trust it, "0": flat buffer, default: check once
    if (comm is None and cuda32 and dist.get_backend() == "nccl" and manager is not None and want != "0"
            and _coalesced_state != "off" and all(v is not None for v in views)):
        # The coalescing group is a private torch API applied to transposed views, and no build round has had two GPUs to
        # run it on.  So the FIRST step of a process runs both forms: the flat-buffer all-reduce below
        # (the validated path) on a copy, then the in-place group; the ranks agree on the verdict with one more
        # all-reduce.  Equal -> the group serves every later step; different -> a loud warning, the flat result is
        # installed and the flat path serves the rest of the process.
        reference = None
        if _coalesced_state is None and want != "1":
            reference = torch.cat([v.reshape(-1) for v in views])
            _all_reduce_sum(reference)
            reference.div_(world)
        with manager(device=grads[0].device, async_ops=False):
            for v in views:
                dist.all_reduce(v, op=dist.ReduceOp.SUM)
        torch._foreach_div_(views, float(world))
        if reference is None:
            _coalesced_state = "on"
            return
        got = torch.cat([v.reshape(-1) for v in views])
        # sums of W fp32 terms in two association orders: equal to a few ulps of the largest term
        tol = 1e-5 * float(reference.abs().max()) + 1e-30
        bad = ((got - reference).abs() > tol).any().to(torch.float32).reshape(1)
        _all_reduce_sum(bad)
        if float(bad) == 0.0:
            _coalesced_state = "on"
            return
        import warnings
        warnings.warn("pygda_amd.distributed: the coalesced in-place gradient all-reduce disagrees with the flat-buffer "
                      "all-reduce on this stack; using the flat buffer from here on (PYGDA_AMD_COALESCED_GRADS=0 "
                      "skips the check)")
        _coalesced_state = "off"
        off = 0
        for v in views:
            n = v.numel()
            v.copy_(reference[off:off + n].view_as(v))
            off += n
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    if comm is not None:
        comm.all_reduce_(flat)
    else:
        _all_reduce_sum(flat)
    flat.div_(world)
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
