"""Run the tall GEMM kernels a few times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops
n, k = 150_000, 256
x = torch.randn(n, k, device="cuda"); w = torch.randn(128, k, device="cuda"); gy = torch.randn(n, 128, device="cuda")
for _ in range(10):
    ops.gemm(ops.GEMM_NT, x, w); ops.gemm(ops.GEMM_TN, gy, x)
torch.cuda.synchronize()
