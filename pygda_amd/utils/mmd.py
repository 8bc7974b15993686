"""``MMD`` / ``get_MMD`` / ``guassian_kernel`` (pygda/utils/mmd.py:4-159) on the fused
MI355X kernels (:func:`pygda_amd.ops.mmd_loss`): no ``[n, n, d]`` temporary, recompute in
the backward pass."""
import torch

from .. import distributed
from ..ops import mmd_loss, mmd_loss_rows, sample_rows, selection_csr_host


def guassian_kernel(source, target, kernel_mul=2.0, kernel_num=5, fix_sigma=None):
    """The ``[m, m]`` multi-bandwidth kernel matrix itself (mmd.py:4-55), for callers that
    want it.  Off the training path (the trainers never materialise it): evaluated with
    device tensor ops in row blocks, same arithmetic as the reference."""
    total = torch.cat([source, target], dim=0)
    m = total.size(0)
    rows = [((total.unsqueeze(0) - total[s:s + 256].unsqueeze(1)) ** 2).sum(2) for s in range(0, m, 256)]
    L2 = torch.cat(rows, dim=0)
    bandwidth = fix_sigma if fix_sigma else (torch.sum(L2.data) + 1e-6) / (m ** 2 - m)
    bandwidth = bandwidth / (kernel_mul ** (kernel_num // 2))
    return sum(torch.exp(-L2 / (bandwidth * (kernel_mul ** i))) for i in range(kernel_num))


def get_MMD(source_feat, target_feat, kernel_mul=2.0, kernel_num=5, fix_sigma=None):
    """mean(XX + YY - XY - YX) over the given rows (mmd.py:57-107)."""
    return mmd_loss(source_feat, target_feat, None, None, kernel_mul, kernel_num, fix_sigma)


# Optional source of the row samples: a callable (n_src, n_tgt, times, sampling_num) -> two device
# int64 tensors.  The hipGraph-captured training step installs one that hands out STATIC device
# buffers which it refills (from the same CPU-generator draws) before every replay.
# (source map, target map) or None: full-batch loaders that train on a degree-ordered relabelling of a large power-law
# graph (pygda_amd/data.py::auto_reorder) -- the draws below are made in the CALLER's node numbering, exactly as the
# reference makes them (mmd.py:148-149), and mapped to the rows those nodes occupy in the relabelled batch
# The maps belong to ONE trainer's loaders: BaseGDA installs them for the duration of its own epoch loop
# (`scoped_row_maps` around `_train_epochs`) and nothing stays installed between fits, so a direct MMD() call, a
# graph-mode trainer or another live model never sees a map built for somebody else's graph.
row_maps = None


class scoped_row_maps:
    """``with scoped_row_maps(maps):`` -- install a trainer's (source map, target map) pair for the block and
    restore whatever was installed before (normally None) on the way out, exceptions included."""

    def __init__(self, maps):
        self.maps = maps if maps is not None and any(m is not None for m in maps) else None

    def __enter__(self):
        global row_maps
        self.prev, row_maps = row_maps, self.maps
        return self

    def __exit__(self, *exc):
        global row_maps
        row_maps = self.prev
        return False


def apply_row_maps(source_sample, target_sample, ns=None, nt=None):
    """In place: node ids of the caller's numbering -> rows of the (possibly relabelled) batch.  ``ns`` / ``nt``:
    the row counts the draws were made for; a map of another length belongs to another graph and is refused."""
    if row_maps is not None:
        for sample, m, n in ((source_sample, row_maps[0], ns), (target_sample, row_maps[1], nt)):
            if m is not None:
                if n is not None and m.numel() != n:
                    raise RuntimeError(f"utils.mmd.row_maps: a map of {m.numel()} rows is installed but the MMD "
                                       f"draws are over {n} rows -- the map belongs to another trainer's loader")
                sample.copy_(m[sample])


sample_provider = None
dp_index_provider = None       # data-parallel branch: (n_src, n_tgt, times, per) -> (idx_s, idx_t, sel_s, sel_t)


# ---- the draws of the NEXT MMD() call, prepared beside the forward pass (eager sampled training) --------------------
# MMD() draws its row samples from the CPU generator, builds the selection CSRs of their gradient scatter (a counting
# sort over the batch's ~1.6e5 rows per domain) and ships 1.4 MB through a pinned block: 0.43 ms of host time per
# cfg-S step, on the thread that also has to enqueue the step's ~60 launches -- and with the two branches on two
# streams the step is host-bound.  A trainer that knows the row counts early (the batches' node counts, at the top of
# forward_model) calls prefetch_samples(): a helper thread makes the SAME draws (nothing else reads the CPU generator
# between the two points), the native counting sort and the staging copy release the GIL, and MMD() takes the finished
# block.  Any mismatch (other counts, another device, a call in between) falls back to drawing in place.
_prefetched = None
PREFETCH = __import__("os").environ.get("PYGDA_AMD_MMD_PREFETCH", "1") == "1"


_worker = None          # (thread, job queue): ONE helper thread for the process -- a thread per step cost the training
                        # thread a thread creation (~0.1 ms) per step


def _worker_loop(jobs):
    while True:
        job = jobs.get()
        if job is None:
            return
        ns, nt, times, sampling_num, dev, stream, box, done, stacked = job
        try:
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                s_cpu = torch.randint(ns, (times, sampling_num))
                t_cpu = torch.randint(nt, (times, sampling_num))
                apply_row_maps(s_cpu, t_cpu, ns, nt)
                from ..ops import mmd_samples_to_device
                box["out"] = mmd_samples_to_device(s_cpu, t_cpu, ns, nt, torch.device(dev), stacked=stacked)
        except BaseException as exc:          # surfaced by the consumer
            box["err"] = exc
        finally:
            done.set()


def prefetch_samples(ns, nt, dev, sampling_num=1000, times=5):
    global _prefetched, _worker
    if (not PREFETCH or _prefetched is not None or sample_provider is not None or dp_index_provider is not None
            or torch.device(dev).type != "cuda"):
        return
    stacked = True
    if distributed.active():
        # data-parallel: this rank's 1/W share of every resample, scattered out of its own [times, per, d] block (the
        # same two draws MMD()'s data-parallel branch makes, in the same order)
        sampling_num = -(-sampling_num // distributed.info()["world_size"])
        stacked = False
    import queue
    import threading
    if _worker is None or not _worker[0].is_alive():
        jobs = queue.SimpleQueue()
        th = threading.Thread(target=_worker_loop, args=(jobs,), daemon=True, name="pygda-amd-mmd-draws")
        th.start()
        _worker = (th, jobs)
    stream = torch.cuda.current_stream(dev)
    box, done = {}, threading.Event()
    _worker[1].put((ns, nt, times, sampling_num, dev, stream, box, done, stacked))
    _prefetched = (ns, nt, times, sampling_num, str(torch.device(dev)), done, box)


def _take_prefetched(ns, nt, times, sampling_num, dev):
    global _prefetched
    hit, _prefetched = _prefetched, None
    if hit is None:
        return None
    hit[5].wait()
    if "err" in hit[6]:
        raise hit[6]["err"]
    if hit[:5] != (ns, nt, times, sampling_num, str(torch.device(dev))):
        # left over from a forward pass that never reached its MMD() (an exception in between): its draws are spent,
        # this call makes its own
        import warnings
        warnings.warn("MMD(): discarding row samples prefetched for another call "
                      f"({hit[:4]} against {(ns, nt, times, sampling_num)})")
        return None
    return hit[6]["out"]


def MMD(source_feat, target_feat, sampling_num=1000, times=5, *, scale=1.0, add=None):
    """``scale`` / ``add`` (keyword-only, not in the reference): return ``add + scale * MMD`` from the loss
    kernels themselves -- the trainers' ``loss = CE + MMD(...) * weight`` line without glue kernels.

    Average of ``times`` MMDs over ``sampling_num`` rows drawn with replacement per
    domain (mmd.py:109-159).  The draws come from the CPU default generator exactly as in
    the reference (``torch.randint`` without a device, :148-149), so a seeded run samples
    the same rows; only the 2 x times x sampling_num indices cross PCIe."""
    dev = source_feat.device
    if distributed.active():
        # data-parallel: each rank draws its 1/W share of every resample from ITS mini-batch,
        # the rows are all-gathered (2 x times x sampling_num x d floats) and every rank
        # evaluates the same global-batch MMD; gradients return to the rows a rank owns.
        w = distributed.info()["world_size"]
        per = -(-sampling_num // w)
        ns, nt = source_feat.size(0), target_feat.size(0)
        if dp_index_provider is not None:       # captured step: static device buffers, refilled per replay
            s_idx, t_idx, sel_s, sel_t = dp_index_provider(ns, nt, times, per)
        elif dev.type == "cuda" and (ready := _take_prefetched(ns, nt, times, per, dev)) is not None:
            s_idx, t_idx, sel = ready           # drawn beside the forward passes (prefetch_samples)
            sel_s, sel_t = (sel[0], sel[1], sel[4]), (sel[2], sel[3], sel[4])
        else:
            s_cpu, t_cpu = torch.randint(ns, (times, per)), torch.randint(nt, (times, per))
            apply_row_maps(s_cpu, t_cpu, ns, nt)
            if dev.type == "cuda":      # one pinned block, one asynchronous copy (pageable .to() calls drain the stream)
                from ..ops import mmd_samples_to_device
                s_idx, t_idx, sel = mmd_samples_to_device(s_cpu, t_cpu, ns, nt, dev, stacked=False)
                sel_s, sel_t = (sel[0], sel[1], sel[4]), (sel[2], sel[3], sel[4])
            else:
                ones = torch.ones(times * per, dtype=torch.float32, device=dev)
                sel_s = tuple(t.to(dev) for t in selection_csr_host(s_cpu, ns, 0, per)) + (ones,)
                sel_t = tuple(t.to(dev) for t in selection_csr_host(t_cpu, nt, 0, per)) + (ones,)
                s_idx, t_idx = s_cpu.to(dev), t_cpu.to(dev)
        s_rows = distributed.all_gather_rows(sample_rows(source_feat, s_idx, sel_s))    # [W, times, per, d]
        t_rows = distributed.all_gather_rows(sample_rows(target_feat, t_idx, sel_t))
        d = source_feat.size(1)
        s_rows = s_rows.permute(1, 0, 2, 3).reshape(times, w * per, d)
        t_rows = t_rows.permute(1, 0, 2, 3).reshape(times, w * per, d)
        out = mmd_loss_rows(s_rows, t_rows)
        out = out * scale if scale != 1.0 else out
        return out if add is None else add + out
    if sample_provider is not None:
        s_idx, t_idx, sel = sample_provider(source_feat.size(0), target_feat.size(0), times, sampling_num)
        return mmd_loss(source_feat, target_feat, s_idx, t_idx, sel=sel, scale=scale, add=add)
    ready = _take_prefetched(source_feat.size(0), target_feat.size(0), times, sampling_num, dev)
    if ready is not None:
        s_idx, t_idx, sel = ready
        return mmd_loss(source_feat, target_feat, s_idx, t_idx, sel=sel, scale=scale, add=add)
    source_sample = torch.randint(source_feat.size(0), (times, sampling_num))
    target_sample = torch.randint(target_feat.size(0), (times, sampling_num))
    apply_row_maps(source_sample, target_sample, source_feat.size(0), target_feat.size(0))
    from ..ops import mmd_samples_to_device
    s_idx, t_idx, sel = mmd_samples_to_device(source_sample, target_sample, source_feat.size(0), target_feat.size(0), dev)
    return mmd_loss(source_feat, target_feat, s_idx, t_idx, sel=sel, scale=scale, add=add)
