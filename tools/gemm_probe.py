"""One shape, forward only: us per call of ops.gemm(NT) at 157 k x 128 x 128 (and K = 256)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops
dev = "cuda:0"
def t(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
out = {}
for n, k in ((157000, 128), (157000, 256)):
    x = torch.randn(n, k, device=dev); w = torch.randn(128, k, device=dev); gy = torch.randn(n, 128, device=dev); wn = torch.randn(128, k, device=dev)
    out[f"fwd_{k}"] = round(t(lambda: ops.gemm(ops.GEMM_NT, x, w)), 1)
    out[f"dgrad_{k}"] = round(t(lambda: ops.gemm(ops.GEMM_NN, gy, wn)), 1)
print(os.environ.get("PYGDA_AMD_GEMM_DBG", "0"), os.environ.get("PYGDA_AMD_GEMM_SPLIT_F16", "1"), json.dumps(out))
