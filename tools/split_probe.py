"""Where does a split-graph cfg-A step go?  HIP events around the three graph replays (no profiler)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pygda_amd.models import A2GNN
from pygda_amd.ops import dropout_state
dev = "cuda:0"
src, tgt = bench.make_cfg_a()
m = A2GNN(6775, 128, 5, num_layers=2, lr=0.01, weight_decay=0.005, epoch=60, dropout=0.5, s_pnums=0, t_pnums=10,
          weight=10, device=dev, verbose=0, use_hip_graph=True)
torch.manual_seed(0)
state = m._prepare(src, tgt)
m._train_epochs(*state, epochs=range(10))
g = m._graphed
print(type(g).__name__)
def t(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = {"replay_only_ms": t(g._replay), "refill_plus_replay_ms": t(lambda: (g._refill(), g._replay()))}
if hasattr(g, "graphs"):
    g1, g2, g3 = g.graphs
    out["g1_alone_ms"] = t(g1.replay); out["g2_alone_ms"] = t(g2.replay); out["g3_alone_ms"] = t(g3.replay)
    def host_only():
        t0 = time.perf_counter(); g._replay(); return time.perf_counter() - t0
    torch.cuda.synchronize()
    hs = []
    for _ in range(20):
        torch.cuda.synchronize(); hs.append(host_only())
    out["host_replay_call_us"] = round(sum(hs) / len(hs) * 1e6, 1)
    main = torch.cuda.current_stream()
    E = lambda: torch.cuda.Event(enable_timing=True)
    acc = [0.0] * 4
    for _ in range(50):
        e0, ea, eb, e2, e3 = E(), E(), E(), E(), E()
        torch.cuda.synchronize()
        dropout_state.next_step(torch.device(dev))
        e0.record(main)
        g._sa.wait_stream(main); g._sb.wait_stream(main)
        with torch.cuda.stream(g._sa): g1.replay(); ea.record(g._sa)
        with torch.cuda.stream(g._sb): g2.replay(); eb.record(g._sb)
        main.wait_stream(g._sa); main.wait_stream(g._sb)
        e2.record(main)
        g3.replay(); e3.record(main)
        torch.cuda.synchronize()
        for k, v in enumerate((e0.elapsed_time(ea), e0.elapsed_time(eb), e0.elapsed_time(e2), e2.elapsed_time(e3))):
            acc[k] += v
    out["from_start_to_g1_end_us"] = round(acc[0] / 50 * 1e3, 1); out["to_g2_end_us"] = round(acc[1] / 50 * 1e3, 1)
    out["to_join_us"] = round(acc[2] / 50 * 1e3, 1); out["g3_us"] = round(acc[3] / 50 * 1e3, 1)
else:
    def host_only():
        t0 = time.perf_counter(); g._replay(); return time.perf_counter() - t0
    hs = []
    for _ in range(20):
        torch.cuda.synchronize(); hs.append(host_only())
    out["host_replay_call_us"] = round(sum(hs) / len(hs) * 1e6, 1)
print(json.dumps(out))
