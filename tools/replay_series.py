#!/usr/bin/env python
"""Per-replay host and device times of the captured cfg-A step from its FIRST replay on (the one
slow replay early in a captured fit; it sits inside a 20-step timed region once a capture holds four or eight steps).

    PYGDA_AMD_GRAPH_UNROLL=4 python tools/replay_series.py [steps, default 60] [warm-up steps as in bench.py, default 5]

Prints, per replay in launch order: steps in it, host microseconds inside the launch call (refill + hipGraphLaunch + the
asynchronous read-back), device milliseconds per step between consecutive replay ends."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                   # noqa: E402

import bench                                                                   # noqa: E402
from pygda_amd import hipgraph as _hg                                          # noqa: E402
from pygda_amd.models import A2GNN                                             # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    hp = dict(hid=128, classes=5, L=2, lr=0.01, wd=0.005, dropout=0.5, s_pnums=0, t_pnums=10, weight=10)
    src, tgt = bench.make_cfg_a(seed=200, degrees="uniform")
    model = A2GNN(src.x.size(1), hp["hid"], hp["classes"], num_layers=hp["L"], lr=hp["lr"], weight_decay=hp["wd"],
                  epoch=warm + steps, dropout=hp["dropout"], s_pnums=hp["s_pnums"], t_pnums=hp["t_pnums"],
                  weight=hp["weight"], device=dev, verbose=0)
    torch.manual_seed(1234)
    state = model._prepare(src, tgt)
    rec = []
    orig_multi, orig_single = _hg.GraphedStep.launch_multi, _hg.GraphedStep.launch

    def traced(fn, per):
        def run(self):
            h0 = time.perf_counter()
            t = fn(self)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            rec.append((per(self), h0, time.perf_counter(), ev))
            return t
        return run

    _hg.GraphedStep.launch_multi = traced(orig_multi, lambda g: g.unroll)
    _hg.GraphedStep.launch = traced(orig_single, lambda g: 1)
    t0 = time.perf_counter()
    model._train_epochs(*state, epochs=range(warm))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n_warm = len(rec)
    model._train_epochs(*state, epochs=range(warm, warm + steps))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"unroll {getattr(model._graphed, 'unroll', 1)}; warm-up {warm} steps in {1e3 * (t1 - t0):.1f} ms (capture included), "
          f"{steps} steps in {1e3 * (t2 - t1):.2f} ms = {1e3 * (t2 - t1) / steps:.4f} ms/step")
    for i, (per, h0, h1, ev) in enumerate(rec):
        dms = rec[i - 1][3].elapsed_time(ev) / per if i and i != n_warm else float("nan")
        print(f"  replay {i:3d}{' (warm-up)' if i < n_warm else '':10s} steps {per}  host {1e6 * (h1 - h0):8.1f} us  "
              f"gap to next call {1e6 * ((rec[i + 1][1] if i + 1 < len(rec) else h1) - h1):8.1f} us  device {dms:7.4f} ms/step")


if __name__ == "__main__":
    main()
