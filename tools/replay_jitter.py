#!/usr/bin/env python
"""Per-replay timing of the captured cfg-A step over a long run (on the GPU box): device time between the ends of
consecutive replays (HIP events) and the host's time per launch; prints percentiles and the slow replays with their
index -- periodic stalls point at the runtime / the host loop, not at the kernels.  python tools/replay_jitter.py [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_cfg_a
from pygda_amd.models import A2GNN
from pygda_amd import hipgraph

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 600
src, tgt = make_cfg_a(seed=200)
model = A2GNN(src.x.size(1), 128, 5, num_layers=2, lr=0.01, weight_decay=0.005, epoch=epochs + 10, dropout=0.5, s_pnums=0,
              t_pnums=10, weight=10, device="cuda:0", verbose=0)
torch.manual_seed(1234)
state = model._prepare(src, tgt)
model._train_epochs(*state, epochs=range(10))
torch.cuda.synchronize()
rec = []
orig = hipgraph.GraphedStep.launch_multi


def traced(self):
    h0 = time.perf_counter()
    t = orig(self)
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    rec.append((h0, time.perf_counter(), ev))
    return t


hipgraph.GraphedStep.launch_multi = traced
t0 = time.perf_counter()
model._train_epochs(*state, epochs=range(10, 10 + epochs))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
dev = [rec[i - 1][2].elapsed_time(rec[i][2]) for i in range(1, len(rec))]
host = [1e3 * (b - a) for a, b, _ in rec]
gap = [1e3 * (rec[i][0] - rec[i - 1][1]) for i in range(1, len(rec))]       # host time between two launches (report + wait)
q = lambda v, p: sorted(v)[min(len(v) - 1, int(p * len(v)))]
print(f"{epochs} epochs in {1e3 * dt:.1f} ms = {1e3 * dt / epochs:.4f} ms/step; {len(rec)} replays")
print("device ms between replay ends  p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f" % tuple(q(dev, p) for p in (0.1, 0.5, 0.9, 0.99, 0.9999)))
print("host ms per launch call        p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f" % tuple(q(host, p) for p in (0.1, 0.5, 0.9, 0.99, 0.9999)))
print("host ms between launches       p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f" % tuple(q(gap, p) for p in (0.1, 0.5, 0.9, 0.99, 0.9999)))
med = q(dev, 0.5)
slow = [(i, round(d, 3), round(host[i], 3), round(gap[i - 1], 3)) for i, d in enumerate(dev, 1) if d > 1.5 * med]
print("replays slower than 1.5 x median (index, device ms, host launch ms, host gap before ms):", slow[:40], "count", len(slow))
