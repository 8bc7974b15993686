"""``GRADE`` trainer (pygda/models/grade.py:18-300): GCN stack whose concatenated layer
outputs feed a domain loss -- 'JS' (gradient-reversed linear discriminator, fused kernel),
'MMD' (fused sampled MMD kernel at width hid*L + C) or 'C' (label-conditional)."""
import numpy as np
import torch
import torch.nn.functional as F

from ..ops import source_ce
from ..nn import GRADEBase
from ..ops import grl_disc_ce
from ..utils import MMD
from .base import BaseGDA


class GRADE(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, mode='node', num_layers=2, dropout=0., act=F.relu,
                 disc='JS', weight=0.01, weight_decay=0.01, lr=0.001, epoch=200, device='cuda:0',
                 batch_size=0, num_neigh=-1, verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        self.disc, self.weight, self.mode = disc, weight, mode

    def init_model(self, **kwargs):
        return GRADEBase(in_dim=self.in_dim, hid_dim=self.hid_dim, num_classes=self.num_classes,
                         num_layers=self.num_layers, dropout=self.dropout, act=self.act, disc=self.disc,
                         mode=self.mode, **kwargs).to(self.device)

    def forward_model(self, source_data, target_data, alpha):
        net = self.grade
        both = self._stacked_pair(source_data, target_data) if net.training else None
        if both is not None:
            # full-batch node mode: the two passes :162-163 over one network as ONE pass over the block-diagonal pair
            from ..ops import split_rows
            logits, feats = net(both)
            source_logits, target_logits = split_rows(logits, both.ns)
            source_feats, target_feats = split_rows(feats, both.ns)
        else:
            source_logits, source_feats = net(source_data)
            target_logits, target_feats = net(target_data)
        loss = source_ce(source_logits, source_data.y)
        lin = net.discriminator[0]
        domain_loss = 0
        if self.disc == 'JS':                                                      # :169-176
            domain_loss = grl_disc_ce(source_feats, target_feats, lin.weight, lin.bias, alpha)
        elif self.disc == 'MMD':                                                   # :177-182
            # graph mode: len(batch) in the reference = the number of graphs (PyG's Batch.__len__); the pooled feature
            # matrix has one row per graph, so the row counts say the same
            mind = min(source_feats.size(0), target_feats.size(0)) if self.mode == 'graph' else \
                min(source_data.x.size(0), target_data.x.size(0))
            domain_loss = MMD(source_feats[:mind], target_feats[:mind])
        elif self.disc == 'C':                                                     # :183-193
            ratio = 8
            s_l_f = torch.cat([source_feats, ratio * net.one_hot_embedding(source_data.y)], dim=1)
            t_l_f = torch.cat([target_feats, ratio * F.softmax(target_logits, dim=1)], dim=1)
            domain_loss = grl_disc_ce(s_l_f, t_l_f, lin.weight, lin.bias, alpha)
        return loss + domain_loss * self.weight, source_logits, target_logits

    def fit(self, source_data, target_data):
        self._loaders(source_data, target_data)                                    # :219-252
        self.grade = self.init_model(**self.kwargs)
        if torch.device(self.device).type == "cuda":       # one capturable multi-tensor launch (pygda_amd/optim.py)
            from ..optim import Adam
        else:
            Adam = torch.optim.Adam
        optimizer = Adam(self.grade.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        # the step replays as a hipGraph: its only per-epoch scalar, the GRL alpha, is a device tensor there
        # (graph mode re-collates a shuffled batch every epoch: nothing static to capture)
        self._graph_safe_step, self._graph_uses_scalars = self.mode == 'node', self.disc != 'MMD'

        def step(src, tgt, alpha, epoch):
            loss, source_logits, _ = self.forward_model(src, tgt, alpha)
            return loss, source_logits

        self._train_epochs(self.grade, optimizer, step,
                           lambda e: 2 / (1 + np.exp(-10 * e / self.epoch)) - 1)   # :260

    def process_graph(self, data):
        pass

    def predict(self, data, source=False):
        self.grade.eval()
        loader = self.source_loader if source else self.target_loader
        return self._predict_loader(loader, lambda batch: self.grade(batch)[0])
