"""``StruRW`` trainer (pygda/models/strurw.py:21-758), modes ``erm`` / ``mmd`` / ``adv`` on the GS and
GCN re-weighting backbones: every ``ew_freq`` epochs (from ``ew_start`` on) the source edges are
re-weighted by the ratio of class-pair edge probabilities target(pseudo labels) / source(labels).

The reference forms those probabilities through dense ``N x N`` adjacencies on the host
(``to_dense_adj`` + scipy products, strurw.py:508-546) and assigns the weights with ``C^2``
``np.in1d`` passes over the edge list (:476-481); here they are two ``bincount``s over the edge list and
one gather, on the device.

``mode='mixup'`` (strurw.py:259-313) trains ``MixupBase``: per step ``lam ~ Beta(4, 4)`` and a node shuffle from
numpy's global generator (in the reference's order, so a seeded run draws the same ones); the shuffled graph is
handed over as a :class:`~pygda_amd.nn.ShuffledEdges` record -- a renumbering of the source graph, which the
backbone turns into a row permutation of ONE aggregation per layer (pygda_amd/nn/mixup_base.py)."""
import itertools
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..ops import dropout_state, source_ce
from ..metrics import eval_micro_f1
from ..nn.reverse_layer import GradReverse
from ..nn.mixup_base import MixupBase, ShuffledEdges
from ..nn.reweight_gnn import ReweightGNN
from ..utils import MMD, logger
from .base import BaseGDA
from ..nn.linear import DenseLinear


class StruRW(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=2, cls_dim=128, cls_layers=2, dropout=0.,
                 gnn='GS', pooling='mean', reweight=True, pseudo=True, ew_start=100, ew_freq=20, lamb=0.8,
                 mode='erm', act=F.relu, bn=False, weight_decay=0.0001, lr=0.05, epoch=100, device='cuda:0',
                 batch_size=0, num_neigh=-1, verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        assert mode in ['erm', 'mixup', 'mmd', 'adv'], 'unsupport training mode'       # strurw.py:128
        self.gnn, self.lamb, self.mode, self.bn, self.pooling = gnn, lamb, mode, bn, pooling
        self.cls_dim, self.cls_layers, self.reweight = cls_dim, cls_layers, reweight
        self.ew_freq, self.ew_start, self.pseudo = ew_freq, ew_start, pseudo

    def init_model(self, **kwargs):
        if self.mode == 'mixup':                                                        # :163-172
            return MixupBase(in_dim=self.in_dim, hid_dim=self.hid_dim, num_classes=self.num_classes,
                             num_layers=self.num_layers, dropout=self.dropout, rw_lmda=self.lamb,
                             **kwargs).to(self.device)
        return ReweightGNN(input_dim=self.in_dim, gnn_dim=self.hid_dim, output_dim=self.num_classes,
                           cls_dim=self.cls_dim, gnn_layers=self.num_layers, cls_layers=self.cls_layers,
                           backbone=self.gnn, pooling=self.pooling, dropout=self.dropout, bn=self.bn,
                           rw_lmda=self.lamb, **kwargs).to(self.device)

    # `self.gnn` is the backbone NAME until fit() replaces it by the model, as in the reference
    def forward_model(self, source_data, target_data, alpha, epoch):
        target_feat, target_logits = self.gnn.forward(target_data, target_data.x)
        target_pred = torch.max(F.softmax(target_logits, dim=1), dim=1)[1]
        if self.reweight and (epoch + 1) >= self.ew_start:                              # :226-232
            if self.pseudo:
                if (epoch + 1) % self.ew_freq == 0:
                    self.cal_reweight(source_data, target_data, target_pred)
            elif epoch == self.ew_start - 1:
                self.cal_reweight(source_data, target_data, target_pred)
        source_feat, source_logits = self.gnn.forward(source_data, source_data.x)
        loss = source_ce(source_logits, source_data.y)
        if self.mode == 'adv':                                                          # :240-250
            source_dlogits = self.domain_discriminator(GradReverse.apply(source_feat, alpha))
            target_dlogits = self.domain_discriminator(GradReverse.apply(target_feat, alpha))
            domain_label = torch.cat([torch.zeros(source_data.x.shape[0], dtype=torch.long),
                                      torch.ones(target_data.x.shape[0], dtype=torch.long)]).to(source_dlogits.device)
            loss = loss + F.cross_entropy(torch.cat([source_dlogits, target_dlogits], 0), domain_label)
        elif self.mode == 'mmd':                                                        # :251-254
            loss = loss + MMD(source_feat, target_feat)
        return loss, source_logits, target_logits

    def forward_model_mixup(self, source_data, target_data, epoch):
        """:259-313.  The target pass is the un-mixed network (``lam = 1``, identity shuffle)."""
        n_t = target_data.x.shape[0]
        target_feat = self.gnn.feat_bottleneck(target_data.x, target_data.edge_index, target_data.edge_index, 1,
                                               self._arange(n_t), target_data.edge_weight)
        target_logits = self.gnn.feat_classifier(target_feat)
        target_pred = torch.max(F.softmax(target_logits, dim=1), dim=1)[1]
        if self.reweight and (epoch + 1) >= self.ew_start:                              # :281-287
            if self.pseudo:
                if (epoch + 1) % self.ew_freq == 0:
                    self.cal_reweight(source_data, target_data, target_pred)
            elif epoch == self.ew_start - 1:
                self.cal_reweight(source_data, target_data, target_pred)
        lam = np.random.beta(4.0, 4.0)                                                  # :292
        data_b, id_new_value_old = self.shuffle_data(source_data)                       # :293
        source_feat = self.gnn.feat_bottleneck(source_data.x, source_data.edge_index, data_b.edge_index, lam,
                                               id_new_value_old, source_data.edge_weight)
        source_logits = self.gnn.feat_classifier(source_feat)
        loss = source_ce(source_logits, source_data.y)                                  # :311: source labels only
        return loss, source_logits, target_logits

    def _arange(self, n):
        """``np.arange(n)`` of :264 / :696, one object per size (the backbone recognises the identity)."""
        cache = self.__dict__.setdefault("_aranges", {})
        if n not in cache:
            cache[n] = np.arange(n)
        return cache[n]

    def shuffle_data(self, data):
        """:702-733 -- one ``np.random.shuffle`` of ``arange(N)``; the shuffled copy of the graph carries the
        permuted labels and, instead of a renumbered edge tensor (:735-758), the record of how it was derived."""
        id_new_value_old = np.arange(data.x.shape[0])
        np.random.shuffle(id_new_value_old)
        from ..data import Data
        perm = torch.from_numpy(id_new_value_old).to(data.y.device)
        data_b = Data(x=None, edge_index=ShuffledEdges(data.edge_index, id_new_value_old), y=data.y[perm])
        data_b.edge_weight = getattr(data, "edge_weight", None)
        return data_b, id_new_value_old

    def cal_edge_prob_sep(self, src_graph, tgt_graph, tgt_pred):
        """(source, target-by-pseudo-label, target-by-label) class-pair edge probabilities (:489-547):
        ``#edges(c1 -> c2) / (n_c1 * n_c2)`` -- counted on the edge list (duplicates included, as
        ``to_dense_adj`` sums them), float64."""
        c = self.num_classes

        def prob(ei, lab, eps):
            cnt = torch.bincount(lab[ei[0]] * c + lab[ei[1]], minlength=c * c).view(c, c).double()
            per = torch.bincount(lab, minlength=c).double()
            return cnt / (per.view(-1, 1) * per.view(1, -1) + eps)

        return (prob(src_graph.edge_index, src_graph.y, 0.0), prob(tgt_graph.edge_index, tgt_pred, 1e-12),
                prob(tgt_graph.edge_index, tgt_graph.y, 0.0))

    def cal_reweight(self, source_data, target_data, target_pred):
        """:446-487: ``w(u, v) = ratio[label(v), label(u)]``, ratio = target / source probability with
        inf / nan -> 1; replaces ``source_data.edge_weight``."""
        src_prob, tgt_prob, _ = self.cal_edge_prob_sep(source_data, target_data, target_pred)
        ratio = tgt_prob / src_prob
        ratio[torch.isinf(ratio)] = 1
        ratio[torch.isnan(ratio)] = 1
        lab, ei = source_data.y, source_data.edge_index
        source_data.edge_weight = ratio[lab[ei[1]], lab[ei[0]]].float()

    def fit(self, source_data, target_data):
        self._node_loaders(source_data, target_data)
        if not (self.source_loader.full_batch and self.target_loader.full_batch):
            raise NotImplementedError(
                "StruRW with sampled mini-batches: the re-weighting writes `edge_weight` of the batch object it is "
                "handed (strurw.py:487), which a sampled batch does not carry back to the graph")
        self.gnn = self.init_model(**self.kwargs)
        if self.mode == 'adv':
            self.domain_discriminator = DenseLinear(self.hid_dim, 2).to(self.device)
            params = itertools.chain(self.gnn.parameters(), self.domain_discriminator.parameters())
        else:
            params = self.gnn.parameters()
        optimizer = torch.optim.Adam(params, lr=self.lr, weight_decay=self.weight_decay)
        src = next(iter(self.source_loader)).to(self.device)
        tgt = next(iter(self.target_loader)).to(self.device)
        for d in (src, tgt):                                                            # :358-361
            if getattr(d, "edge_weight", None) is None:
                d.edge_weight = torch.ones(d.edge_index.shape[1], device=self.device)
        self._device_graphs = (src, tgt)
        start_time = time.time()
        for epoch in range(self.epoch):
            alpha = 2. / (1. + np.exp(-10. * float(epoch) / self.epoch)) - 1
            self.gnn.train()
            if self.mode == 'mixup':
                if src.x.is_cuda:
                    dropout_state.next_step(src.x.device)      # fresh keep-bits for the fused epilogue's dropout
                loss, _, _ = self.forward_model_mixup(src, tgt, epoch)
            else:
                loss, _, _ = self.forward_model(src, tgt, alpha, epoch)
            epoch_loss = loss.item()
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            logits, labels = self.predict(src)                                          # :421-426, eval mode
            acc = eval_micro_f1(labels, logits.argmax(dim=1))
            secs = time.time() - start_time
            logger(epoch=epoch, loss=epoch_loss, source_train_acc=acc, time=secs, verbose=self.verbose, train=True)
            if self.epoch_hook is not None:
                self.epoch_hook(epoch, epoch_loss, acc, secs)
        source_data.edge_weight = src.edge_weight.to(source_data.edge_index.device)    # the caller's object sees the weights

    def process_graph(self, data):
        pass

    def predict(self, data):
        """Encodes the ``data`` it is given (:669-700)."""
        self.gnn.eval()
        data = data.to(self.device)
        if self.mode == 'mixup':                                                        # :694-696: unit weights, always
            unit = self.__dict__.setdefault("_unit_weights", {})
            key = (data.edge_index.data_ptr(), data.edge_index.shape[1])
            if key not in unit:                            # one tensor per graph: the backbone's CSR cache hits on it
                unit[key] = (data.edge_index, torch.ones(data.edge_index.shape[1], device=data.edge_index.device))
            data.edge_weight = unit[key][1]
            with torch.no_grad():
                logits = self.gnn(data.x, data.edge_index, data.edge_index, 1, self._arange(data.x.shape[0]),
                                  data.edge_weight)
            return logits, data.y
        if getattr(data, "edge_weight", None) is None:
            data.edge_weight = torch.ones(data.edge_index.shape[1], device=data.edge_index.device)
        with torch.no_grad():
            _, logits = self.gnn(data, data.x)
        return logits, data.y
