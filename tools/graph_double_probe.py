"""Is a forked hipGraph's replay paced by the host enqueuing its nodes one by one while the previous replay of
the SAME executable graph is still running?  One forked graph replayed back to back vs two captures of the same
work replayed alternately."""
import sys, time, json
import torch
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
x = torch.randn(sz, device=dev)


def chain(v, k):
    for _ in range(k):
        v = v * 1.0001 + 0.5
    return v


def capture(forked):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        r = x * 2.0
        if forked:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = chain(r + 1, n)
            a = chain(r, n)
            main.wait_stream(side)
        else:
            a = chain(r, n); b = chain(r + 1, n)
        out = a.sum() + b.sum()
    return g, out


def t(fn, it=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / it * 1e6, 1)


res = {}
for forked in (False, True):
    copies = [capture(forked) for _ in range(4)]
    k = {"i": 0}
    def alt(m):
        def f():
            copies[k["i"] % m][0].replay(); k["i"] += 1
        return f
    tag = "forked" if forked else "serial"
    res[tag + "_one_exec"] = t(alt(1)); res[tag + "_two_execs"] = t(alt(2)); res[tag + "_four_execs"] = t(alt(4))
    def host_only():
        torch.cuda.synchronize(); t0 = time.perf_counter(); copies[0][0].replay(); return time.perf_counter() - t0
    res[tag + "_host_call_us"] = round(sum(host_only() for _ in range(20)) / 20 * 1e6, 1)
print(json.dumps(res))
