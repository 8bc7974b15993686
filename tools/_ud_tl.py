import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pygda_amd
src, tgt = bench.make_cfg_a()
torch.manual_seed(0); np.random.seed(0)
m = pygda_amd.models.UDAGCN(6775, 128, 5, num_layers=2, ppmi=True, adv_dim=40, lr=0.01, device="cuda:0", epoch=30, verbose=0)
m.fit(src, tgt)
torch.cuda.synchronize()
