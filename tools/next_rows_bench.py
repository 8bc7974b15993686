"""Timings for the "next" rows (TDSS / DGSDA) on the cfg-A shapes: the Bernstein filter as chain +
Horner (2K launches) against the reference's evaluation order (K + K(K+1)/2 launches) run on the
SAME aggregation kernel, the Laplacian smoothness kernels on the 2-hop smoothing graph, and
ms/step of TDSS.fit / DGSDA.fit.  One JSON line per measurement."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pygda_amd import ops  # noqa: E402
from pygda_amd.graph import build_csr  # noqa: E402
from pygda_amd.models import DGSDA, TDSS  # noqa: E402
from pygda_amd.models.tdss import _add_remaining_self_loops, two_hop_edges  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def bern_reference_order(graph, x, w, K):
    """dgsda_base.py:132-148 on our kernel: tmp[j] = (I+A)^j x, then L^(i+1) tmp[K-i-1] per i."""
    tmp = [x]
    for _ in range(K):
        tmp.append(ops.spmm_axpby(graph, tmp[-1], 1.0, 1.0))
    out = tmp[K] * w[0]
    for i in range(K):
        y = tmp[K - i - 1]
        for _ in range(i + 1):
            y = ops.spmm_axpby(graph, y, 1.0, -1.0)
        out = out + y * w[i + 1]
    return out


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None       # "TDSS" / "DGSDA": trainers only (for rocprofv3)
    src, tgt = bench.make_cfg_a()
    n = tgt.x.size(0)
    ei = tgt.edge_index.to(DEV)
    graph = build_csr(ei, n, add_self_loops="drop", normalize=True, degree_side="row")
    out = []
    for K in (() if only else (8, 15)):
        from math import comb
        coefs = torch.tensor([comb(K, k) / 2 ** K for k in range(K + 1)], device=DEV)
        temp = torch.rand(K + 1, device=DEV, requires_grad=True)
        x = torch.randn(n, 128, device=DEV, requires_grad=True)
        w = (torch.relu(temp) * coefs).detach()
        ref = bern_reference_order(graph, x.detach(), w, K)
        got = ops.bern_filter(x, temp, graph, coefs)
        err = float((got.detach() - ref).abs().max() / ref.abs().max())
        t_fwd = timed(lambda: ops.bern_filter(x.detach(), temp.detach(), graph, coefs))
        t_ref = timed(lambda: bern_reference_order(graph, x.detach(), w, K))

        def fb():
            o = ops.bern_filter(x, temp, graph, coefs)
            o.backward(torch.ones_like(o))
            x.grad = None; temp.grad = None
        t_fb = timed(fb)
        out.append(dict(what="bern_filter", K=K, N=n, d=128, nnz=graph.nnz, launches_fwd=2 * K,
                        launches_reference_order=K + K * (K + 1) // 2, ms_fwd=round(t_fwd, 4),
                        ms_fwd_reference_order=round(t_ref, 4), ms_fwd_bwd=round(t_fb, 4), max_rel_diff=err))
    # Laplacian smoothness on the 2-hop smoothing graph of the target
    if only:
        smooth = None
    else:
      smooth = _add_remaining_self_loops(two_hop_edges(tgt.edge_index, n, 1), n).to(DEV)
    if smooth is not None:
        f = torch.randn(n, 128, device=DEV, requires_grad=True)
        ops.laplacian_loss(f, smooth)

        def lap():
            loss = ops.laplacian_loss(f, smooth)
            loss.backward()
            f.grad = None
        e = smooth.size(1)
        out.append(dict(what="laplacian_loss fwd+bwd", N=n, d=128, E_smooth=int(e), ms=round(timed(lap), 4),
                        reference_temporaries_bytes=4 * e * 128 * 4))
    # trainers, ms per epoch (full batch => one step)
    for name, ctor in (("TDSS", lambda g: TDSS(6775, 128, 5, smooth_mode='K-hop', k=2, t_pnums=10, dropout=0.1,
                                                 device=DEV, epoch=30, verbose=0, use_hip_graph=g)),
                       ("DGSDA", lambda g: DGSDA(6775, 128, 5, K=8, dropout=0.1, device=DEV, epoch=30, verbose=0,
                                                   use_hip_graph=g))):
        for graphed in (False, True):
            if only and (name != only or not graphed):
                continue
            m = ctor(graphed)
            stamps = []
            m.epoch_hook = lambda e_, loss, acc, secs: (torch.cuda.synchronize(), stamps.append(time.perf_counter()))
            torch.manual_seed(0)
            m.fit(src, tgt)
            ms = (stamps[-1] - stamps[9]) / (len(stamps) - 10) * 1e3
            out.append(dict(what=f"{name}.fit cfg-A shapes", hip_graph=graphed, ms_per_step=round(ms, 3)))
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
