"""Concurrency of NARROW long kernels (K-step LDS kernel at d=8: 8 workgroups, ~25 us each) across streams:
(a) one single-stream graph with both chains, (b) one graph with a fork, (c) two graphs on two streams."""
import sys, time, json, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pygda_amd import ops
from pygda_amd.graph import build_csr
dev = "cuda:0"
_, tgt = bench.make_cfg_a()
G = build_csr(tgt.edge_index.to(dev), tgt.num_nodes); G.static = True
n = 5
x = torch.randn(tgt.num_nodes, 8, device=dev); y = torch.randn(tgt.num_nodes, 8, device=dev)
def chain(v):
    for _ in range(n): v = ops.spmm_kstep(G, v, 10)
    return v
chain(x); chain(y); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1): chain(x)
with torch.cuda.stream(s2): chain(y)
torch.cuda.synchronize()
g_serial, g_fork, ga, gb = (torch.cuda.CUDAGraph() for _ in range(4))
with torch.cuda.graph(g_serial):
    a = chain(x); b = chain(y)
with torch.cuda.graph(g_fork):
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2): b2 = chain(y)
    a2 = chain(x)
    main.wait_stream(s2)
with torch.cuda.graph(ga, stream=s1): a3 = chain(x)
with torch.cuda.graph(gb, stream=s2): b3 = chain(y)
torch.cuda.synchronize()
def both():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): ga.replay()
    with torch.cuda.stream(s2): gb.replay()
    main.wait_stream(s1); main.wait_stream(s2)
def eager2():
    with torch.cuda.stream(s1): chain(x)
    with torch.cuda.stream(s2): chain(y)
def t(fn, it=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / it * 1e6, 1)
print(json.dumps({"serial_graph": t(g_serial.replay), "forked_graph": t(g_fork.replay), "chain_a_graph": t(ga.replay),
                  "two_graphs_two_streams": t(both), "eager_two_streams": t(eager2)}))
