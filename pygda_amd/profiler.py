"""Optional HIP-event timing of the path's kernels (off by default: zero overhead).

bench.py switches it on for the timed region to get, per kernel family, the number of
launches, their summed duration on the launch stream, and the algorithmic bytes / flops
they moved -- the inputs of the roofline line.  Events are recorded on the stream the
kernels are launched on (torch's current stream, which is what the C ABI is handed)."""
from collections import defaultdict

import torch

enabled = False
_records = defaultdict(list)       # name -> [(start, end, launches, bytes, flops, extra)]


def start():
    global enabled
    _records.clear()
    enabled = True


def stop():
    global enabled
    enabled = False


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class _Region:
    __slots__ = ("name", "launches", "nbytes", "flops", "extra", "s")

    def __init__(self, name, launches, nbytes, flops, extra):
        self.name, self.launches, self.nbytes, self.flops, self.extra = name, launches, nbytes, flops, extra

    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True)
        self.s.record()

    def __exit__(self, *exc):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _records[self.name].append((self.s, e, self.launches, self.nbytes, self.flops, self.extra))
        return False


_NULL = _Null()


def region(name, launches=1, nbytes=0, flops=0, **extra):
    """Context manager bracketing a kernel family's launches with HIP events (a shared no-op object while the
    profiler is off: the call sites sit on the eager training loop's host path).  ``extra``: further per-call byte
    counts of the kernel family (summed by :func:`summary`), e.g. the bytes a kernel really moves over HBM next to
    the algorithmic bytes it stands for."""
    if not enabled:
        return _NULL
    return _Region(name, launches, nbytes, flops, extra)


def summary():
    """name -> dict(calls, launches, ms, avg_us, bytes, flops); call after a device sync."""
    out = {}
    for name, recs in _records.items():
        ms = sum(s.elapsed_time(e) for s, e, *_ in recs)
        launches = sum(r[2] for r in recs)
        out[name] = dict(calls=len(recs), launches=launches, ms=ms,
                         avg_us=1e3 * ms / max(launches, 1),
                         bytes=sum(r[3] for r in recs), flops=sum(r[4] for r in recs))
        for r in recs:
            for k, v in r[5].items():
                out[name][k] = out[name].get(k, 0) + v
    return out
