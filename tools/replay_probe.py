"""Where does a cfg-A epoch go?  GPU time of the captured step alone (back-to-back replays, no host
work), replay + per-step refill, and the full fit() epoch (refill + replay + loss/accuracy read-back)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pygda_amd.models import A2GNN

dev = "cuda:0"
src, tgt = bench.make_cfg_a()
m = A2GNN(6775, 128, 5, num_layers=2, lr=0.01, weight_decay=0.005, epoch=60, dropout=0.5, s_pnums=0, t_pnums=10,
          weight=10, device=dev, verbose=0, use_hip_graph=True)
torch.manual_seed(0)
if os.environ.get("PROBE_NO_TLOGITS"):
    m.compute_target_logits = False
if os.environ.get("PROBE_NO_MMD"):
    import pygda_amd.models.a2gnn as _a
    _a.MMD = lambda s, t: (s.sum() + t.sum()) * 0.0
if os.environ.get("PROBE_NO_OVERLAP"):
    m.overlap_streams = False
state = m._prepare(src, tgt)
m._train_epochs(*state, epochs=range(10))
if os.environ.get("PROBE_NO_TLOGITS"):
    pass
g = m._graphed
def t(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = dict(replay_only_ms=t(lambda: g.graph.replay()), refill_plus_replay_ms=t(lambda: g()),
           refill_host_only_ms=None)
t0 = time.perf_counter()
for _ in range(200): g._refill()
out["refill_host_only_ms"] = (time.perf_counter() - t0) / 200 * 1e3
torch.cuda.synchronize()
t0 = time.perf_counter(); m._train_epochs(*state, epochs=range(10, 60)); torch.cuda.synchronize()
out["fit_epoch_ms"] = (time.perf_counter() - t0) / 50 * 1e3
print(json.dumps(out))
