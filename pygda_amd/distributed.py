"""Data-parallel plumbing over ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
the GPU box, ``gloo`` in the CPU tests).  One process per GPU; graph and features are
replicated; ranks own disjoint seed mini-batches (``NeighborLoader(rank, world_size)``).
Two exchange steps per training step, both small and latency-bound:

* one flat all-reduce of every gradient (averaged over ranks);
* an all-gather of the MMD sample rows, so that the pairwise loss is taken over the GLOBAL
  batch (the cross-domain statistic cannot be formed from per-rank losses).
"""
import torch
import torch.distributed as dist


def info():
    if dist.is_available() and dist.is_initialized():
        return dict(rank=dist.get_rank(), world_size=dist.get_world_size())
    return dict(rank=0, world_size=1)


def active():
    """True when the data-parallel exchange steps must run.  ``PYGDA_AMD_FORCE_DP=1`` switches them
    on for a 1-rank group too (single-GPU test of the multi-GPU code path)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PYGDA_AMD_FORCE_DP") == "1"


def allreduce_grads(params):
    """ONE collective per step: flatten, all-reduce(sum), divide by the world size."""
    if not active():
        return
    params = list(params)
    for p in params:                      # a rank whose batch never touched p still joins
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
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
        if dist.get_backend() == "nccl":       # single-buffer form: no staging copies, graph-capturable
            dist.all_gather_into_tensor(out, x.contiguous())
        else:
            dist.all_gather(list(out.unbind(0)), x.contiguous())
        ctx.rank, ctx.world = dist.get_rank(), w
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank] * float(ctx.world)


def all_gather_rows(x):
    return _AllGatherRows.apply(x) if active() else x.unsqueeze(0)
