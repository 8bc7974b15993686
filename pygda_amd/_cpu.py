"""Host threads and the container's CPU quota.

PyTorch sizes its intra-op pool (OpenMP / MKL) by the CPUs it can SEE -- 128 on the MI355X boxes -- while the container is
allowed 16 cores' worth of time per 100 ms (cgroup ``cpu.max`` = ``1600000 100000``).  One parallel region on the host
(feature generation before a fit, the CPU oracle of a bench line) wakes 128 threads that work and then spin; the group
exceeds its quota within ~12 ms and the kernel freezes EVERY thread of the process for the rest of the period.  Measured
on the GPU box (``/sys/fs/cgroup/cpu.stat``, round 5): 1 - 3 throttle events per cfg-S bench process and 4 per cfg-A
process with PyTorch's pool, none with the pool capped (`host_cpu.cgroup_throttle_events_during_this_process` in every
bench line).  What the cap does NOT explain (same session): the rare 30 - 50 ms host stall inside a kernel launch of a
sampled step and the slow replays of a captured fit still occur with zero throttle events.

``respect_cpu_quota()`` caps the pool at the quota (minus a few cores for the training thread, the loaders' producer
threads and the runtime's own; divided by the ranks of the node) once per process, at import of the package.
``PYGDA_AMD_CPU_THREADS`` = a number forces the pool size, ``0`` leaves PyTorch's choice alone."""
import os

import torch

_done = False


def cpu_quota():
    """Cores' worth of CPU time the cgroup allows (float), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2
            quota, period = fh.read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:                                                               # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            quota = float(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = float(fh.read())
        return quota / period if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def throttle_counters():
    """(nr_throttled, throttled_usec) of this cgroup so far, or None: what a bench line reports as a delta."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            with open(path) as fh:
                kv = dict(line.split()[:2] for line in fh if line.strip())
            if "nr_throttled" in kv:
                return int(kv["nr_throttled"]), int(kv.get("throttled_usec", kv.get("throttled_time", 0)))
        except (OSError, ValueError):
            continue
    return None


def pool_size_for(quota, visible, reserve=4):
    """Threads for the intra-op pool under a quota of ``quota`` cores with ``visible`` CPUs in the affinity mask: the quota
    less ``reserve`` cores for the threads that must never wait (at least half of it, at least 1), never more than
    PyTorch would take."""
    if quota is None:
        return visible
    room = int(quota) - reserve
    return max(1, min(visible, max(room, int(quota) // 2)))


def respect_cpu_quota():
    """-> the pool size in force after the call (idempotent)."""
    global _done
    env = os.environ.get("PYGDA_AMD_CPU_THREADS", "auto")
    if _done or env == "0":
        return torch.get_num_threads()
    _done = True
    cur = torch.get_num_threads()
    if env != "auto":
        want = max(1, int(env))
    else:
        quota = cpu_quota()
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))          # one process per GPU: the ranks of a node share it
        want = pool_size_for(None if quota is None else quota / ranks, cur)
    if want < cur:
        torch.set_num_threads(want)
    return torch.get_num_threads()
