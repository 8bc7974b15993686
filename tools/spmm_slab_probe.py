"""HBM-regime aggregation, measured alternatives: the same kernel on COLUMN SLABS of the
feature matrix (a slab of 2 M nodes x 16 columns = 128 MB fits the 256 MB Infinity Cache; the index arrays are then
re-read once per slab), on a degree-sorted relabelling of an R-MAT graph, and plain -- time per full aggregation.
JSON lines on stdout."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd import _lib, ops                      # noqa: E402
from pygda_amd.graph import build_csr                # noqa: E402
from tools.spmm_sweep import rmat_edges              # noqa: E402


def timed(fn, iters=5):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def slab_spmm(G, x, y, width):
    """y = A x, `width` columns at a time through gda_spmm_csr_f32 (ldx = ldy = d: slabs are strided views)."""
    L = _lib.lib()
    n, d = x.shape
    for c in range(0, d, width):
        _lib.check(L.gda_spmm_csr_f32(_lib.ptr(G.rowptr), _lib.ptr(G.colidx), _lib.ptr(G.val), n, width,
                                      x.data_ptr() + 4 * c, d, y.data_ptr() + 4 * c, d, None, _lib.stream()), "gda_spmm_csr_f32")


def case(name, n, ei, d=128):
    G = build_csr(ei, n, validate=False)
    nnz = G.nnz
    x = torch.randn(n, d, device="cuda")
    y = torch.empty_like(x)
    alg = nnz * 8 + (n + 1) * 4 + 2 * n * d * 4
    out = {"case": name, "N": n, "nnz": nnz, "d": d, "algorithmic_GB": round(alg / 1e9, 3)}
    us = timed(lambda: ops.spmm_kstep(G, x, 1))
    out["plain_us"] = round(us, 1); out["plain_alg_GBs"] = round(alg / us / 1e3, 1)
    ref = ops.spmm_kstep(G, x, 1)
    for w in (64, 32, 16, 8):
        us = timed(lambda: slab_spmm(G, x, y, w))
        out[f"slab{w}_us"] = round(us, 1)
        assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4), w      # (hub rows: chunked in `plain`, sequential in the slabs)
    print(json.dumps(out), flush=True)
    del G, x, y, ref
    torch.cuda.empty_cache()


def main():
    gen = torch.Generator(device="cuda").manual_seed(200)
    for n in (1_000_000, 2_000_000, 5_000_000):
        half = n * 10
        a = torch.randint(0, n, (half,), generator=gen, device="cuda")
        b = torch.randint(0, n, (half,), generator=gen, device="cuda")
        case(f"uniform-{n // 1_000_000}M", n, torch.stack([torch.cat([a, b]), torch.cat([b, a])]))
    ei = rmat_edges(22, 32_000_000, gen)
    ei = torch.cat([ei, ei.flip(0)], dim=1)
    n = 1 << 22
    case("rmat-2^22", n, ei)
    # degree-sorted relabelling: hubs first, so the rows every row gathers most often share cache lines / pages
    deg = torch.bincount(ei[1], minlength=n)
    order = torch.argsort(deg, descending=True)
    new_id = torch.empty_like(order)
    new_id[order] = torch.arange(n, device="cuda")
    case("rmat-2^22-degree-sorted", n, new_id[ei])


if __name__ == "__main__":
    main()
