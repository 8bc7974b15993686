"""``DANE`` trainer (pygda/models/dane.py:21-621), node mode and graph mode (pooled embeddings, no skip-gram
term, linear classifier -- :171-176, 219-229, 323-331, 448-456, 492-493): shared GNN encoder trained as an
LSGAN generator (5 discriminator updates per generator update) with a skip-gram edge loss
(degree^0.75 negative sampling) and source cross-entropy.

Every random draw is made on the host generator.  The reference draws some of them on
whatever device the tensors live on (``torch.multinomial`` of device weights, :382), which on
its CPU path is the CPU generator -- so a seeded run here reproduces the reference's CPU run.
As in AdaGCN, the encoder outputs are detached inside the discriminator loop: the reference
back-propagates into the encoder there and discards the result (``g_optimizer.zero_grad()``
at :511 precedes the only generator step)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..data import to_undirected
from ..nn import GNNBase
from ..nn.a2gnn_base import global_mean_pool
from .base import BaseGDA


import os
from ..nn.linear import DenseLinear

FUSED_LSGAN = os.environ.get("PYGDA_AMD_FUSED_LSGAN", "1") == "1"


class DANE(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers, mode='node', dropout=0., gnn='gcn', k=5,
                 train_mode='unsup', tgt_rate=0.05, act=F.relu, weight_decay=1e-5, lr=0.001, epoch=200,
                 device='cuda:0', batch_size=0, num_neigh=-1, verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        assert train_mode in ['semi', 'unsup'], 'unsupport training mode'
        self.gnn_arc, self.k, self.train_mode, self.tgt_rate, self.mode = gnn, k, train_mode, tgt_rate, mode

    def init_model(self, **kwargs):
        return GNNBase(in_dim=self.in_dim, hid_dim=self.hid_dim, num_classes=self.num_classes,
                       num_layers=self.num_layers, dropout=self.dropout, gnn=self.gnn_arc, mode=self.mode,
                       **kwargs).to(self.device)

    # -- host-side draws (CPU generator), moved to the device as index tensors -----------
    def _draw(self, n, count, replacement):
        return torch.multinomial(torch.ones(n), count, replacement=replacement).to(self.device)

    def _lsgan_rows(self, es, et):
        i_s = self._draw(es.shape[0], 8 * self.sample_size, True)
        i_t = self._draw(et.shape[0], 8 * self.sample_size, True)
        return self.domain_discriminator(es[i_s]), self.domain_discriminator(et[i_t])

    def _lsgan_loss(self, es, et, target_s, target_t):
        """``((D(es[i_s]) - target_s) ** 2).mean() + ((D(et[i_t]) - target_t) ** 2).mean()`` over 8 x sample_size
        sampled rows per domain (:339-350 with targets (0, 1), :468-470 with (1, 0)); the draws are the reference's.
        On the GPU the discriminator the trainer builds (Linear - ReLU - Linear(., 1)) runs as one GEMM plus the fused
        LSGAN head per domain (ops.lsgan_head); anything else composes it."""
        i_s = self._draw(es.shape[0], 8 * self.sample_size, True)
        i_t = self._draw(et.shape[0], 8 * self.sample_size, True)
        xs, xt = es[i_s], et[i_t]
        D = self.domain_discriminator
        from ..ops import lsgan_head, lsgan_head_ok
        if (FUSED_LSGAN and len(D) == 3 and isinstance(D[0], nn.Linear) and isinstance(D[1], nn.ReLU)
                and isinstance(D[2], nn.Linear) and lsgan_head_ok(xs, D[0].weight, D[2].weight)
                and lsgan_head_ok(xt, D[0].weight, D[2].weight)):
            par = (D[0].weight, D[0].bias, D[2].weight, D[2].bias)
            return lsgan_head(xs, *par, target_s) + lsgan_head(xt, *par, target_t)
        return ((D(xs) - target_s) ** 2).mean() + ((D(xt) - target_t) ** 2).mean()

    def forward_model(self, source_data, target_data):
        for _ in range(5):
            discriminator_loss = self.train_d(source_data, target_data)
        generator_loss = self.train_g(source_data, target_data)
        graph = self.mode == 'graph'                                                    # :171-176
        source_logits = self.gnn(source_data.x, source_data.edge_index, batch=source_data.batch if graph else None)
        target_logits = self.gnn(target_data.x, target_data.edge_index, batch=target_data.batch if graph else None)
        return discriminator_loss + generator_loss, source_logits, target_logits

    def _embed(self, data):
        """Encoder output; graph mode: mean-pooled per graph (:323-331, :448-456)."""
        e = self.gnn.feat_bottleneck(data.x, data.edge_index)
        return global_mean_pool(e, data.batch) if self.mode == 'graph' else e

    def train_d(self, source_data, target_data):                                        # :301-355
        self.gnn.eval()
        with torch.no_grad():
            es, et = self._embed(source_data), self._embed(target_data)
        self.d_optimizer.zero_grad()
        loss = self._lsgan_loss(es, et, 0.0, 1.0)
        loss.backward()
        self.d_optimizer.step()
        return loss.item()

    def L_GCN(self, embedding, nodes_weight, idx_u, idx_v, k):                          # :357-389
        eu, ev = embedding[idx_u], embedding[idx_v]
        neg = [embedding[torch.multinomial(nodes_weight, self.sample_size, replacement=False).to(embedding.device)]
               for _ in range(k)]
        loss = -torch.sum(F.logsigmoid(torch.sum(eu * ev, dim=1)))
        for i in range(k):
            loss = loss - torch.sum(F.logsigmoid(torch.sum(eu * neg[i] * (-1), dim=1)))
        return loss

    def L_cluster(self, labelsA, embA, labelsB, embB):                                  # :391-424
        loss = 0.0
        for i in range(self.num_classes):
            a, b = labelsA == i, labelsB == i
            if bool(a.any()) and bool(b.any()):
                loss = loss + torch.sum((embA[a].mean(0) - embB[b].mean(0)) ** 2)
        return loss / self.num_classes

    def _pick_edges(self, data):
        """sample_size positive edges + the degree^0.75 negative-sampling weights (host draws)."""
        ei_cpu = data.edge_index.cpu()
        pick = torch.multinomial(torch.ones(ei_cpu.shape[1]), self.sample_size, replacement=False)
        w = torch.pow(torch.unique(ei_cpu[0], return_counts=True)[1], 0.75)                 # host: see header
        return w, ei_cpu[0][pick].to(self.device), ei_cpu[1][pick].to(self.device)

    def train_g(self, source_data, target_data):                                        # :426-516
        self.gnn.train()
        es = self._embed(source_data)
        out_s = self.gnn.feat_classifier(es, source_data.edge_index)
        et = self._embed(target_data)
        out_t = self.gnn.feat_classifier(et, target_data.edge_index)
        l_adv = self._lsgan_loss(es, et, 1.0, 0.0)
        if self.mode == 'node':
            edges_s, edges_t = self._pick_edges(source_data), self._pick_edges(target_data)   # draw order of :480-487
            l_gcn = self.L_GCN(es, *edges_s, self.k) + self.L_GCN(et, *edges_t, self.k)
        else:
            l_gcn = 0                                                                   # :492-493: no skip-gram term
        l_ce = F.cross_entropy(out_s, source_data.y)
        if self.train_mode == 'semi':
            n_t = et.shape[0]
            lab = self._draw(n_t, int(self.tgt_rate * n_t), False)
            l_ce = l_ce + F.cross_entropy(out_t[lab], target_data.y[lab])
            loss = l_gcn + l_adv * 0.1 + l_ce + self.L_cluster(source_data.y, es, target_data.y[lab], et[lab])
        else:
            loss = l_gcn + l_ce + l_adv * 0.1
        self.g_optimizer.zero_grad()
        loss.backward()
        from .base import _allreduce_grads
        _allreduce_grads(self.g_optimizer)
        self.g_optimizer.step()
        return loss.item()

    def fit(self, source_data, target_data):
        import time
        from ..metrics import eval_micro_f1
        from ..utils import logger
        if self.mode == 'node':
            for d in (source_data, target_data):                                        # :203-207
                if not d.is_undirected():
                    d.edge_index = to_undirected(d.edge_index, d.num_nodes)
            self.sample_size = min(source_data.x.shape[0], target_data.x.shape[0])
        elif self.mode == 'graph':
            self.sample_size = min(len(source_data), len(target_data))                  # :220: number of graphs
        self._loaders(source_data, target_data)
        self.gnn = self.init_model(**self.kwargs)
        self.domain_discriminator = nn.Sequential(DenseLinear(self.hid_dim, self.hid_dim), nn.ReLU(),
                                                  DenseLinear(self.hid_dim, 1)).to(self.device)
        self.g_optimizer = torch.optim.Adam(self.gnn.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        self.d_optimizer = torch.optim.Adam(self.domain_discriminator.parameters(), lr=self.lr,
                                            weight_decay=self.weight_decay)
        start = time.time()
        for epoch in range(self.epoch):
            epoch_loss, logits, labels = 0, [], []
            for src, tgt in zip(self.source_loader, self.target_loader):
                self.gnn.train()
                src, tgt = src.to(self.device), tgt.to(self.device)
                loss, source_logits, _ = self.forward_model(src, tgt)
                epoch_loss += loss
                logits.append(source_logits.detach()); labels.append(src.y)
            acc = eval_micro_f1(torch.cat(labels), torch.cat(logits).argmax(dim=1))
            secs = time.time() - start
            logger(epoch=epoch, loss=epoch_loss, source_train_acc=acc, time=secs, verbose=self.verbose, train=True)
            if self.epoch_hook is not None:
                self.epoch_hook(epoch, epoch_loss, acc, secs)

    def process_graph(self, data):
        pass

    def predict(self, data, source=False):
        self.gnn.eval()
        loader = self.source_loader if source else self.target_loader
        graph = self.mode == 'graph'                                                    # :561-564
        return self._predict_loader(loader, lambda b: self.gnn(b.x, b.edge_index, batch=b.batch if graph else None))
