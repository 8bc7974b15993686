"""Tall-skinny GEMM: the hand-written MFMA kernels (ops.gemm) against the BLAS, forward / dgrad / wgrad,
over the row counts of cfg-A (full batch) and cfg-S (sampled sub-graphs).  One JSON line per shape."""
import json, os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops

dev = "cuda:0"


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def blas_wgrad(gy, x, s=32):
    n = x.size(0)
    rows = (n // s) * s
    gw = torch.bmm(gy[:rows].reshape(s, n // s, -1).transpose(1, 2), x[:rows].reshape(s, n // s, -1)).sum(0)
    return gw if rows == n else gw + gy[rows:].t() @ x[rows:]


for n in (9360, 40000, 75000, 150000, 157000, 300000):
    for k in (128, 256):
        x = torch.randn(n, k, device=dev); w = torch.randn(128, k, device=dev); gy = torch.randn(n, 128, device=dev)
        out = dict(N=n, K=k, out=128,
                   fwd_ours=t(lambda: ops.gemm(ops.GEMM_NT, x, w)), fwd_blas=t(lambda: F.linear(x, w)),
                   dgrad_ours=t(lambda: ops.gemm(ops.GEMM_NN, gy, w)), dgrad_blas=t(lambda: gy @ w),
                   wgrad_ours=t(lambda: ops.gemm(ops.GEMM_TN, gy, x)), wgrad_blas=t(lambda: blas_wgrad(gy, x)))
        gf = 2.0 * n * k * 128 / 1e3            # MFLOP per product -> TF = gf / us / 1e3
        out.update(fwd_ours_frac=gf / out["fwd_ours"] / 1e3 / 157.3, dgrad_ours_frac=gf / out["dgrad_ours"] / 1e3 / 157.3,
                   wgrad_ours_frac=gf / out["wgrad_ours"] / 1e3 / 157.3, fwd_blas_frac=gf / out["fwd_blas"] / 1e3 / 157.3)
        print(json.dumps({a: (round(b, 3 if a.endswith("frac") else 1) if isinstance(b, float) else b) for a, b in out.items()}), flush=True)
