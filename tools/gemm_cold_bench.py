"""The tall GEMM kernels with COLD operands: inside a cfg-S step every projection reads activations another kernel has
just written (80-160 MB per operand: no reuse across calls), while tools/gemm_bench.py repeats one call on one set of
buffers, which the 256 MB Infinity Cache then serves.  Here every call takes the next of `sets` operand sets (default
8 x ~240 MB), so each operand comes from HBM.  One JSON line per shape: us per call, warm (one set) and cold."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops

dev = "cuda:0"
sets = int(os.environ.get("SETS", "8"))


def t(fns, reps=48):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, k in ((157000, 128), (157000, 256)):
    xs = [torch.randn(n, k, device=dev) for _ in range(sets)]
    gys = [torch.randn(n, 128, device=dev) for _ in range(sets)]
    w = torch.randn(128, k, device=dev)
    mk = lambda f: [(lambda i=i: f(i)) for i in range(sets)]
    fwd, dgrad, wgrad = (mk(lambda i: ops.gemm(ops.GEMM_NT, xs[i], w)), mk(lambda i: ops.gemm(ops.GEMM_NN, gys[i], w)),
                         mk(lambda i: ops.gemm(ops.GEMM_TN, gys[i], xs[i])))
    out = dict(N=n, K=k, out=128, sets=sets, slabs_env=os.environ.get("PYGDA_AMD_WGRAD_SLABS"),
               fwd_warm=t(fwd[:1]), fwd_cold=t(fwd), dgrad_warm=t(dgrad[:1]), dgrad_cold=t(dgrad),
               wgrad_warm=t(wgrad[:1]), wgrad_cold=t(wgrad))
    gf = 2.0 * n * k * 128 / 1e3
    out.update({a.replace("_cold", "_cold_frac"): gf / b / 1e3 / 157.3 for a, b in out.items() if a.endswith("_cold")})
    print(json.dumps({a: (round(b, 3 if a.endswith("frac") else 1) if isinstance(b, float) else b) for a, b in out.items()}), flush=True)
    del xs, gys
    torch.cuda.empty_cache()
