#!/usr/bin/env python
"""Micro-benchmark of the MMD kernels at the A2GNN shapes (times=5, n=1000, d=128) on the path the trainer takes
(host-drawn samples, selection CSRs, fused segment-reduce + scatter); run under rocprofv3 --kernel-trace --stats
for per-kernel durations, or alone for the fwd+bwd time per call (HIP events)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops

torch.manual_seed(0)
s = torch.randn(9360, 128, device="cuda").relu().requires_grad_()
t = (torch.randn(5484, 128, device="cuda") + 0.2).relu().requires_grad_()
si, ti, sel = ops.mmd_samples_to_device(torch.randint(0, 9360, (5, 1000)), torch.randint(0, 5484, (5, 1000)), 9360, 5484,
                                        torch.device("cuda"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(5):
    ops.mmd_loss(s, t, si, ti, sel=sel).backward()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    loss = ops.mmd_loss(s, t, si, ti, sel=sel)
    loss.backward()
b.record()
torch.cuda.synchronize()
print(f"loss {float(loss):.6f} grad {float(s.grad.abs().sum()):.6e} {float(t.grad.abs().sum()):.6e} "
      f"fwd+bwd {1e3 * a.elapsed_time(b) / reps:.1f} us/call (eager launches)")
