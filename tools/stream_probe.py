"""How fast does this box stream the tall weight gradient's operands?  Plain torch reductions / copies over the same
[158720, 128] + [158720, 128|256] fp32 arrays (warm: the second pass finds them in the Infinity Cache when they fit)."""
import json, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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


n = 158720
out = {}
for k in (128, 256):
    gy, x = torch.randn(n, 128, device=dev), torch.randn(n, k, device=dev)
    mb = (gy.numel() + x.numel()) * 4 / 1e6
    s_us = t(lambda: (gy.sum(), x.sum()))
    c_us = t(lambda: gy.clone())
    out[f"k={k}"] = {"MB": mb, "two_sums_us": round(s_us, 1), "two_sums_TBs": round(mb / s_us, 2),
                     "clone_gy_us": round(c_us, 1), "clone_TBs_rw": round(2 * gy.numel() * 4 / 1e6 / c_us, 2)}
print(json.dumps(out))
