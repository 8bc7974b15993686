#!/usr/bin/env python
"""Epoch time of the other BASELINE.json configurations' trainers at the cfg-A stand-in shapes (full batch):
GRADE (configs[2]), UDAGCN / AdaGCN (configs[3]), eager and with the default captured step.

    python tools/other_configs_bench.py [model ...]     # JSON lines; models: grade_mmd grade_js udagcn adagcn
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                  # noqa: E402
import pygda_amd                              # noqa: E402

DEV = "cuda:0"
EPOCHS = int(os.environ.get("EPOCHS", "40"))


def build(name, graphed):
    kw = dict(device=DEV, epoch=EPOCHS, verbose=0, use_hip_graph=graphed)
    M = pygda_amd.models
    if name == "grade_mmd":
        return M.GRADE(6775, 128, 5, num_layers=5, dropout=0.1, disc="MMD", weight=0.01, lr=0.01, **kw)
    if name == "grade_js":
        return M.GRADE(6775, 128, 5, num_layers=5, dropout=0.1, disc="JS", weight=0.01, lr=0.01, **kw)
    if name == "udagcn":
        return M.UDAGCN(6775, 128, 5, num_layers=2, ppmi=True, adv_dim=40, lr=0.01, **kw)
    if name == "adagcn":
        return M.AdaGCN(6775, 128, 5, num_layers=2, dropout=0.4, adv_dim=40, gp_weight=5, domain_weight=0.1, lr=0.01,
                        weight_decay=0.01, **kw)
    raise SystemExit(f"unknown model {name}")


def main():
    names = sys.argv[1:] or ["grade_mmd", "grade_js", "udagcn", "adagcn"]
    src, tgt = bench.make_cfg_a()
    for name in names:
        for graphed in (False, None):
            torch.manual_seed(0); np.random.seed(0)
            m = build(name, graphed)
            stamps = []
            m.epoch_hook = lambda e, loss, acc, secs: stamps.append((time.perf_counter(), loss))
            t0 = time.perf_counter()
            m.fit(src, tgt)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            # steady state: the last half of the epochs (the first ones hold graph ingestion, PPMI builds, capture)
            half = len(stamps) // 2
            ms = (t1 - stamps[half - 1][0]) / (len(stamps) - half) * 1e3
            print(json.dumps({"model": name, "execution": "eager" if graphed is False else
                              ("hipGraph replay" if getattr(m, "_graphed", None) is not None else "eager (default)"),
                              "ms_per_epoch": round(ms, 3), "first_epochs_incl_setup_ms": round((stamps[half - 1][0] - t0) * 1e3, 1),
                              "finite": bool(np.isfinite(stamps[-1][1]))}), flush=True)
            del m
            import gc
            torch.cuda.synchronize(); gc.collect()


if __name__ == "__main__":
    main()
