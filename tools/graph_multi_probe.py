"""Two independent chains as TWO single-stream hipGraphs replayed concurrently on two streams (event fork /
join between the replays), against one graph holding both chains serially."""
import sys, time, json
import torch
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
x, y = torch.randn(sz, device=dev), torch.randn(sz, device=dev)


def chain(v, k):
    for _ in range(k):
        v = v * 1.0001 + 0.5
    return v


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ga, gb, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(ga, stream=s1):
    a = chain(x, n).sum()
with torch.cuda.graph(gb, stream=s2):
    b = chain(y, n).sum()
with torch.cuda.graph(gs):
    c = chain(x, n).sum() + chain(y, n).sum()
torch.cuda.synchronize()


def both():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()
    main.wait_stream(s1); main.wait_stream(s2)


def t(fn, it=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / it * 1e6, 1)


print(json.dumps({"one_graph_serial": t(gs.replay), "graph_a_alone": t(ga.replay), "two_graphs_two_streams": t(both)}))
