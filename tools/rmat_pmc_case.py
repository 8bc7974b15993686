#!/usr/bin/env python
"""Six full-graph aggregations (d = 128) on the graphs of bench.py's HBM-regime lines -- the uniform 5 M-node / 100 M-edge
graph of bench.hbm_regime_probe, or the R-MAT 2^22 graph of bench.rmat_probe (32 M edges, symmetrised) as generated or
degree-ordered -- the command the PMC passes of tools/profile.sh wrap (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
runs) to put counter traffic beside the algorithmic bytes of those lines.
    python tools/rmat_pmc_case.py uniform|asgen|reorder"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import ops                      # noqa: E402
from pygda_amd.data import degree_order        # noqa: E402
from pygda_amd.graph import build_csr          # noqa: E402
from tools.spmm_sweep import rmat_edges        # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "asgen"
gen = torch.Generator(device="cuda").manual_seed(200)
d = 128
if mode == "uniform":
    n = 5_000_000
    half = n * 20 // 2
    a = torch.randint(0, n, (half,), generator=gen, device="cuda")
    b = torch.randint(0, n, (half,), generator=gen, device="cuda")
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    del a, b
else:
    n = 1 << 22
    ei = rmat_edges(22, 32_000_000, gen)
    ei = torch.cat([ei, ei.flip(0)], dim=1)
if mode == "reorder":
    ei = degree_order(ei, n)[ei]
G = build_csr(ei, n, validate=False)
x = torch.randn(n, d, device="cuda", generator=gen)
for _ in range(6):
    y = ops.spmm_kstep(G, x, 1)
torch.cuda.synchronize()
print(mode, G.nnz, float(y[0, 0]), float(y.double().sum()), float(y.double().abs().sum()))
