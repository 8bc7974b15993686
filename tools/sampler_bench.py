#!/usr/bin/env python
"""Host neighbour-sampler throughput (1 M nodes / 20 M edges, 1024 seeds, fan-out [15, 10])."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd.sampler import NeighborSampler
n, e = 1_000_000, 20_000_000
g = torch.Generator().manual_seed(0)
ei = torch.randint(0, n, (2, e), generator=g)
seeds = torch.randint(0, n, (1024,), generator=g)
print("cpus", os.cpu_count())
for th in (1, 4, 8, 16, 32):
    S = NeighborSampler(ei, n, threads=th)
    S.sample(seeds, [15, 10], seed=0)
    t0 = time.time()
    for i in range(10):
        n_id, sub = S.sample(seeds + i, [15, 10], seed=i)
    print(th, "threads: %.2f ms/batch" % ((time.time() - t0) / 10 * 1e3), n_id.numel(), "nodes", sub.size(1), "edges", flush=True)
