#!/usr/bin/env python
"""Aggregation-kernel sweep: achieved GB/s of gda_spmm_csr_f32 against the algorithmic bytes
(nnz*8 + (N+1)*4 + 2*N*d*4) from the cache-resident citation size up to the HBM-roofline size
of BASELINE.json configs[4] (5 M nodes / 100 M directed edges per domain, d = 128).

    python tools/spmm_sweep.py [--big]        # JSON lines on stdout
"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops                      # noqa: E402
from pygda_amd.graph import build_csr          # noqa: E402


def rmat_edges(n_log2, e, gen, a=0.57, b=0.19, c=0.19):
    """R-MAT (power-law) edge list on 2^n_log2 nodes."""
    src = torch.zeros(e, dtype=torch.int64, device="cuda")
    dst = torch.zeros(e, dtype=torch.int64, device="cuda")
    for _ in range(n_log2):
        r = torch.rand(e, device="cuda", generator=gen)
        right = (r >= a) & (r < a + b) | (r >= a + b + c)        # quadrants b and d: dst bit set
        down = r >= a + b                                         # quadrants c and d: src bit set
        src = src * 2 + down.long()
        dst = dst * 2 + right.long()
    return torch.stack([src, dst])


def run(name, n, ei, d, iters=20, warm=3):
    G = build_csr(ei, n, validate=False)
    nnz = G.nnz
    x = torch.randn(n, d, device="cuda")
    for _ in range(warm):
        ops.spmm_kstep(G, x, 1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        y = ops.spmm_kstep(G, x, 1)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    alg = nnz * 8 + (n + 1) * 4 + 2 * n * d * 4
    gather = nnz * (8 + 4 * d) + n * d * 4
    deg = (G.rowptr[1:] - G.rowptr[:-1])
    print(json.dumps({"case": name, "N": n, "nnz": nnz, "d": d, "max_deg": int(deg.max()), "us": round(us, 2),
                      "algorithmic_GBs": round(alg / us / 1e3, 1), "frac_of_8TBs": round(alg / us / 1e3 / 8000, 4),
                      "gather_model_GBs": round(gather / us / 1e3, 1)}), flush=True)
    del G, x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    gen = torch.Generator(device="cuda").manual_seed(200)
    cases = [("citation-target", 5484, 16234), ("200k/2M", 200_000, 2_000_000), ("1M/20M", 1_000_000, 20_000_000)]
    if a.big:
        cases.append(("5M/100M uniform (configs[4] per-domain)", 5_000_000, 100_000_000))
    for name, n, e in cases:
        ei = torch.randint(0, n, (2, e), generator=gen, device="cuda")
        for d in (128,) if n > 1_000_000 else (128, 5):
            run(name, n, ei, d)
        del ei
    # power-law: R-MAT 2^20 nodes, 16 M edges (and 2^22 / 64 M with --big)
    for lg, e in ((20, 16_000_000),) + (((22, 64_000_000),) if a.big else ()):
        ei = rmat_edges(lg, e, gen)
        run(f"rmat 2^{lg}/{e // 1_000_000}M", 1 << lg, ei, 128)
        del ei


if __name__ == "__main__":
    main()
