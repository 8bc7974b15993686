"""Where a cfg-S step's wall time goes on the host: cProfile of the eager sampled-training loop (bench.run_cfg_s's
step) + GPU-busy time per step from HIP events.  python tools/cfgs_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pygda_amd import ops  # noqa: E402
from pygda_amd.data import NeighborLoader  # noqa: E402
from pygda_amd.models import A2GNN  # noqa: E402
from pygda_amd.models.base import _allreduce_grads  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
if os.environ.get("PYGDA_AMD_NOGC") == "1":          # is a slow step the cyclic collector?
    import gc
    gc.disable()
dev = "cuda:0"
N, B, fan = 5_000_000, 1024, [15, 10]
src = bench.make_cfg_s(N, 20, 256, 5, 200, dev)
tgt = bench.make_cfg_s(N, 20, 256, 5, 201, dev)
model = A2GNN(256, 128, 5, num_layers=2, lr=0.01, weight_decay=0.005, epoch=1, dropout=0.5, s_pnums=0, t_pnums=10,
              weight=10, device=dev, batch_size=B, num_neigh=fan, verbose=0)
torch.manual_seed(1234)
net, optimizer, step_fn, alpha_fn = model._prepare(src, tgt)
g = torch.Generator().manual_seed(7)
need = (steps + 10) * B
kw = dict(device=dev)
sl = NeighborLoader(src, fan, batch_size=B, input_nodes=torch.randint(0, N, (need,), generator=g), **kw)
tl = NeighborLoader(tgt, fan, batch_size=B, input_nodes=torch.randint(0, N, (need,), generator=g), **kw)
it = zip(iter(sl), iter(tl))
t_fetch = t_fwd = t_bwd = t_opt = 0.0


def one_step(timed=False):
    global t_fetch, t_fwd, t_bwd, t_opt
    a = time.perf_counter()
    s, t = next(it)
    b = time.perf_counter()
    ops.dropout_state.next_step(s.x.device)
    net.train()
    loss, _ = step_fn(s, t, 0.0, 0)
    c = time.perf_counter()
    optimizer.zero_grad()
    loss.backward()
    d = time.perf_counter()
    _allreduce_grads(optimizer)
    optimizer.step()
    e = time.perf_counter()
    if timed:
        t_fetch += b - a; t_fwd += c - b; t_bwd += d - c; t_opt += e - d


for _ in range(10):
    one_step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
pr = cProfile.Profile()
t0 = time.perf_counter()
ev[0].record()
pr.enable()
for i in range(steps):
    one_step(True)
    ev[i + 1].record()
pr.disable()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"sampler: {sl.sampler_description()}")
print(f"wall {1e3 * wall / steps:.3f} ms/step; host enqueue per step: fetch {1e3 * t_fetch / steps:.3f} fwd {1e3 * t_fwd / steps:.3f} "
      f"bwd {1e3 * t_bwd / steps:.3f} opt {1e3 * t_opt / steps:.3f} ms")
print("event-to-event ms:", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.2f}" for i in range(steps)))
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(45)
print(sio.getvalue()[:9000])
