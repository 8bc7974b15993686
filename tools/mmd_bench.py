#!/usr/bin/env python
"""Times of the MMD paths at the trainers' shapes (times = 5, 1000 rows per domain drawn from ns / nt feature rows):
forward (gather + pair kernel + finalize) and backward (fold + scatter), HIP events over `reps` calls each.

    python tools/mmd_bench.py [d ...]          default: 128 645
Modes per width: the register-resident one-pass kernel (d in 32..128), the chunked one-pass kernel, the two-pass fp32 kernels."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                   # noqa: E402

from pygda_amd import ops                                                      # noqa: E402


def timed(fn, reps=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    widths = [int(a) for a in sys.argv[1:]] or [128, 645]
    times, n, ns, nt = 5, 1000, 9360, 5484
    gen = torch.Generator().manual_seed(5)
    for d in widths:
        s = torch.randn(ns, d, generator=gen).relu().to(dev).requires_grad_()
        t = (torch.randn(nt, d, generator=gen) * 1.3 + 0.2).relu().to(dev).requires_grad_()
        si, ti = torch.randint(0, ns, (times, n), generator=gen), torch.randint(0, nt, (times, n), generator=gen)
        idx = ops.mmd_samples_to_device(si, ti, ns, nt, dev)
        modes = [("two_pass", False, "0")]
        if d % 32 == 0 and d <= 128:
            modes.append(("one_pass_registers", True, "0"))
        modes.append(("one_pass_chunked", True, "always"))
        for name, one, chunked in modes:
            ops.MMD_ONE_PASS, ops.MMD_CHUNKED = one, chunked
            loss = [None]

            def fwd():
                loss[0] = ops.mmd_loss(s, t, idx[0], idx[1], sel=idx[2])

            def both():
                fwd()
                torch.autograd.grad(loss[0], (s, t))

            f = timed(fwd)
            fb = timed(both)
            print(json.dumps({"d": d, "mode": name, "fwd_us": round(f, 1), "fwd_bwd_us": round(fb, 1), "loss": float(loss[0])}))
        ops.MMD_ONE_PASS, ops.MMD_CHUNKED = True, "auto"


if __name__ == "__main__":
    main()
