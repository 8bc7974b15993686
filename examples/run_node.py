#!/usr/bin/env python
"""Node-level domain-adaptation run with the command line of the reference's benchmark scripts
(benchmark/node/a2gnn.py:23-39 and siblings) and their result line.  ``--source`` / ``--target`` name
citation datasets under ``--data_root`` (ACMv9 / DBLPv7 / Citationv1: ``<name>_docs.txt`` etc.,
pygda/datasets/citation.py); ``--synthetic`` substitutes the shape-identical stand-ins of bench.py
when the files are not at hand.

    python examples/run_node.py --model a2gnn --source ACMv9 --target DBLPv7 --nhid 128 --num_layers 2 \\
        --lr 0.01 --weight_decay 0.005 --epochs 200 --dropout_ratio 0.5 --s_pnums 0 --t_pnums 10 --weight 10
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygda_amd.data import to_undirected                     # noqa: E402
from pygda_amd.datasets import CitationDataset               # noqa: E402
from pygda_amd.metrics import eval_macro_f1, eval_micro_f1   # noqa: E402
from pygda_amd import models                                 # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--model', default='a2gnn', choices=['a2gnn', 'grade', 'udagcn', 'adagcn', 'tdss', 'dgsda', 'gnn'])
p.add_argument('--seed', type=int, default=200)
p.add_argument('--num_layers', type=int, default=3)
p.add_argument('--lr', type=float, default=0.001)
p.add_argument('--weight_decay', type=float, default=0.0)
p.add_argument('--nhid', type=int, default=128)
p.add_argument('--dropout_ratio', type=float, default=0.1)
p.add_argument('--device', default='cuda:0')
p.add_argument('--source', default='ACMv9')
p.add_argument('--target', default='DBLPv7')
p.add_argument('--epochs', type=int, default=800)
p.add_argument('--filename', default='test.txt')
p.add_argument('--adv', action='store_true')
p.add_argument('--weight', type=float, default=0.1)
p.add_argument('--s_pnums', type=int, default=0)
p.add_argument('--t_pnums', type=int, default=20)
p.add_argument('--data_root', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'data', 'Citation'))
p.add_argument('--synthetic', action='store_true', help='shape-identical stand-in graphs (bench.make_cfg_a)')
args = p.parse_args()

torch.manual_seed(args.seed)
np.random.seed(args.seed)
if args.synthetic:
    import bench
    source_data, target_data = bench.make_cfg_a(seed=args.seed)
else:
    source_data = CitationDataset(os.path.join(args.data_root, args.source), args.source)[0]
    target_data = CitationDataset(os.path.join(args.data_root, args.target), args.target)[0]
for d in (source_data, target_data):                         # benchmark/node/a2gnn.py:92-97
    if not d.is_undirected():
        d.edge_index = to_undirected(d.edge_index, d.num_nodes)
num_features = source_data.x.size(1)
num_classes = len(np.unique(source_data.y.cpu().numpy()))
common = dict(in_dim=num_features, hid_dim=args.nhid, num_classes=num_classes, num_layers=args.num_layers,
              weight_decay=args.weight_decay, lr=args.lr, dropout=args.dropout_ratio, epoch=args.epochs,
              device=args.device)
extra = {'a2gnn': dict(weight=args.weight, adv=args.adv, s_pnums=args.s_pnums, t_pnums=args.t_pnums),
         'tdss': dict(smooth_mode='K-hop', s_pnums=args.s_pnums, t_pnums=args.t_pnums),
         'dgsda': {}, 'grade': {}, 'udagcn': {}, 'adagcn': {}, 'gnn': {}}[args.model]
cls = {'a2gnn': models.A2GNN, 'grade': models.GRADE, 'udagcn': models.UDAGCN, 'adagcn': models.AdaGCN,
       'tdss': models.TDSS, 'dgsda': models.DGSDA, 'gnn': models.GNN}[args.model]
model = cls(**common, **extra)
model.fit(source_data, target_data)
logits, labels = model.predict(target_data)
preds = logits.argmax(dim=1)
results = (f'{args.model},source,{args.source},target,{args.target},micro-f1,{eval_micro_f1(labels, preds)},'
           f'macro-f1,{eval_macro_f1(labels, preds)},auc,0.0')
with open(args.filename, 'a+') as f:
    f.write(results + '\n')
print(results)
