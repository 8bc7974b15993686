"""Optional HIP-event timing of the path's kernels (off by default: zero overhead).

bench.py switches it on for the timed region to get, per kernel family, the number of
launches, their summed duration on the launch stream, and the algorithmic bytes / flops
they moved -- the inputs of the roofline line.  Events are recorded on the stream the
kernels are launched on (torch's current stream, which is what the C ABI is handed)."""
from collections import defaultdict
from contextlib import contextmanager

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


@contextmanager
def region(name, launches=1, nbytes=0, flops=0, **extra):
    """``extra``: further per-call byte counts of the kernel family (summed by :func:`summary`), e.g. the bytes a
    kernel really moves over HBM next to the algorithmic bytes it stands for."""
    if not enabled:
        yield
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    try:
        yield
    finally:
        e.record()
        _records[name].append((s, e, launches, nbytes, flops, extra))


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
