"""Host cost of hipGraphLaunch by graph topology (ROCm 7.0 / torch 2.10): the same 40 small kernels captured
(a) on one stream, (b) with ONE kernel forked to a side stream at the root, (c) with a two-way fork over the last
two kernels only, (d) as two chains of 20.  Per case: host time inside replay() with an idle device, and the
period of back-to-back replays.  One JSON line."""
import json
import time

import torch

dev = "cuda:0"
x = torch.randn(16384, device=dev)


def chain(v, k):
    for _ in range(k):
        v = v * 1.0001 + 0.5
    return v


def capture(kind, n=40):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if kind == "serial":
            out = chain(x, n)
        elif kind == "one_side_kernel_at_root":
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = x + 1.0
            a = chain(x, n - 2)
            main.wait_stream(side)
            out = a + b
        elif kind == "fork_last_two":
            a = chain(x, n - 3)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = a + 1.0
            c = a * 2.0
            main.wait_stream(side)
            out = b + c
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = chain(x, n // 2 - 1)
            a = chain(x, n // 2 - 1)
            main.wait_stream(side)
            out = a + b
    return g, out


res = {}
for kind in ("serial", "one_side_kernel_at_root", "fork_last_two", "two_chains"):
    g, out = capture(kind)
    for _ in range(5):
        g.replay()
    host = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    res[kind] = {"host_call_us": round(1e6 * sorted(host)[len(host) // 2], 1), "period_us": round((time.perf_counter() - t0) / 200 * 1e6, 1)}
print(json.dumps(res))
