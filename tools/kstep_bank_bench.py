"""A/B of the K-step kernel's LDS placement (csrc/gda_kstep.hip) at the cfg-A target graph: the same plan with node i
at word 32 + i (flags 0) and with the bank-aware placement (flags 1), K = 1 and K = 10, d = 128, column-major
operands, HIP events around back-to-back launches.  Prints one JSON line per case; results must be bit-identical."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from pygda_amd import _lib, graph as G_


def main():
    dev = torch.device("cuda:0")
    _, tgt = bench.make_cfg_a(feat=8)
    n = tgt.x.shape[0]
    g = G_.build_csr(tgt.edge_index.to(dev), n)
    L = _lib.lib()
    nnz = g.nnz
    d, n_pad = 128, (n + 3) // 4 * 4
    x = torch.randn(d, n_pad, device=dev)
    out = {}
    for transposed in (False, True):
        rp, ci, va = (g.t_rowptr, g.t_colidx, g.t_val) if transposed else (g.rowptr, g.colidx, g.val)
        rp_h, ci_h, va_h = rp.cpu().numpy(), ci[:nnz].cpu().numpy(), va[:nnz].cpu().numpy()
        for flags in (0, 1):
            cap = L.gda_kstep_plan_bytes(12)
            buf = torch.empty(cap, dtype=torch.uint8)
            S = L.gda_kstep_plan_host_ex(rp_h.ctypes.data, ci_h.ctypes.data, va_h.ctypes.data, n, flags, buf.data_ptr(), cap)
            plan = buf[:L.gda_kstep_plan_bytes(S)].to(dev)
            for K in (0, 1, 10):
                y = torch.empty_like(x)
                run = lambda: _lib.check(L.gda_kstep_lds_colmajor_f32(_lib.ptr(plan), S, n, d, K, _lib.ptr(x), n_pad, _lib.ptr(y),
                                                                     n_pad, None, None, _lib.stream()), "kstep")
                for _ in range(20):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1000 / 200
                key = (transposed, K)
                if key in out:
                    assert torch.equal(out[key][:, :n], y[:, :n]), "placement changed the result"
                out[key] = y
                print(json.dumps({"transposed": transposed, "bank_aware": flags, "slots": S, "K": K, "us_per_launch": round(us, 2),
                                  "n": n, "nnz": nnz, "d": d}), flush=True)


if __name__ == "__main__":
    main()
