#!/usr/bin/env python
"""Micro-benchmark of the MMD kernels at the A2GNN shapes (times=5, n=1000, d=128); run under
rocprofv3 --kernel-trace --stats for per-kernel durations."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops

torch.manual_seed(0)
s = torch.randn(9360, 128, device="cuda").relu().requires_grad_()
t = (torch.randn(5484, 128, device="cuda") + 0.2).relu().requires_grad_()
si = torch.randint(0, 9360, (5, 1000), device="cuda")
ti = torch.randint(0, 5484, (5, 1000), device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    loss = ops.mmd_loss(s, t, si, ti)
    loss.backward()
torch.cuda.synchronize()
print(float(loss))
