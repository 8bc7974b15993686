"""How does hipGraph replay schedule two independent branches?  A root kernel, then chain A (n kernels) and
chain B (n kernels) forked onto two streams and joined; replay time against one chain alone, for the
orders in which the branches can be ISSUED during capture (side stream first / main stream first / both on
side streams)."""
import sys, time, json
import torch

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sz = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 20)
xs = [torch.randn(sz, device=dev) for _ in range(4)]


def chain(x, k):
    for _ in range(k):
        x = x * 1.0001 + 0.5
    return x


def capture(mode):
    side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        r = xs[0] * 2.0
        if mode == "single":
            a = chain(r, n)
            out = a.sum()
        elif mode == "serial2":
            a = chain(r, n); b = chain(r + 1, n)
            out = a.sum() + b.sum()
        elif mode == "side_first":
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = chain(r + 1, n)
            a = chain(r, n)
            main.wait_stream(side)
            out = a.sum() + b.sum()
        elif mode == "main_first":
            ev = torch.cuda.Event(); ev.record(main)
            a = chain(r, n)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                b = chain(r + 1, n)
            main.wait_stream(side)
            out = a.sum() + b.sum()
        elif mode == "both_side":
            side.wait_stream(main); side2.wait_stream(main)
            with torch.cuda.stream(side):
                b = chain(r + 1, n)
            with torch.cuda.stream(side2):
                a = chain(r, n)
            main.wait_stream(side); main.wait_stream(side2)
            out = a.sum() + b.sum()
        elif mode == "interleaved":
            side.wait_stream(main)
            a, b = r, r + 1
            for _ in range(n):
                with torch.cuda.stream(side):
                    b = b * 1.0001 + 0.5
                a = a * 1.0001 + 0.5
            main.wait_stream(side)
            out = a.sum() + b.sum()
    return g, out


res = {}
for mode in ("single", "serial2", "side_first", "main_first", "both_side", "interleaved"):
    g, out = capture(mode)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    res[mode] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
print(json.dumps(res))
