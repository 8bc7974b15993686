#!/usr/bin/env python
"""DANE step time at the cfg-A stand-in shapes with the fused LSGAN discriminator head on / off
(PYGDA_AMD_FUSED_LSGAN): one JSON line per setting (ms per epoch over the last half of the epochs)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                  # noqa: E402
import pygda_amd                              # noqa: E402
from pygda_amd.models import dane as dane_mod  # noqa: E402

EPOCHS = int(os.environ.get("EPOCHS", "16"))


def main():
    src, tgt = bench.make_cfg_a()
    for d in (src, tgt):          # DANE draws sample_size nodes without replacement from the nodes that HAVE out-edges
        n = d.x.shape[0]          # (dane.py:382): the stand-in graphs have isolated nodes, so close a ring over all nodes
        ring = torch.stack([torch.arange(n), (torch.arange(n) + 1) % n])
        d.edge_index = torch.cat([d.edge_index, ring, ring.flip(0)], dim=1)
    for fused in (True, False):
        dane_mod.FUSED_LSGAN = fused
        torch.manual_seed(0); np.random.seed(0)
        m = pygda_amd.models.DANE(6775, 128, 5, num_layers=2, dropout=0.1, gnn='gcn', lr=0.001, device="cuda:0",
                                  epoch=EPOCHS, verbose=0)
        stamps = []
        m.epoch_hook = lambda e, loss, acc, secs: stamps.append((time.perf_counter(), float(loss)))
        m.fit(src, tgt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        half = len(stamps) // 2
        print(json.dumps({"model": "dane", "fused_lsgan_head": fused, "epochs": EPOCHS,
                          "ms_per_epoch": round((t1 - stamps[half - 1][0]) / (len(stamps) - half) * 1e3, 3),
                          "finite": bool(np.isfinite(stamps[-1][1]))}), flush=True)
        del m


if __name__ == "__main__":
    main()
