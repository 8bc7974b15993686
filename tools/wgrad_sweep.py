"""k_tall_wgrad at cfg-S's shapes for different slab counts (PYGDA_AMD_WGRAD_SLABS, read once per process: run one process
per value):  python tools/wgrad_sweep.py   -> one JSON line."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops

dev = "cuda:0"


def t(fn, reps=40):
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


out = {"slabs": os.environ.get("PYGDA_AMD_WGRAD_SLABS", "default")}
for n in (158720, 317440):
    for k in (128, 256):
        x = torch.randn(n, k, device=dev)
        gy = torch.randn(n, 128, device=dev)
        ref = (gy.double().t() @ x.double())
        got = ops.gemm(ops.GEMM_TN, gy, x)
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        out[f"{n}x128x{k}"] = {"us": round(t(lambda: ops.gemm(ops.GEMM_TN, gy, x)), 1), "rel_err": err}
print(json.dumps(out))
