"""``GNN`` trainer (pygda/models/gnn.py:17-245): no adaptation -- cross-entropy on the source
graph only, evaluated on the target.  ``predict(data)`` takes the graph itself (unlike the
GDA trainers it stores no loaders)."""
import torch
import torch.nn.functional as F

from ..ops import source_ce
from ..data import NeighborLoader
from ..metrics import eval_micro_f1
from ..nn import GNNBase
from ..utils import logger
from .base import BaseGDA, _allreduce_grads


class GNN(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=2, dropout=0., gnn='gcn', act=F.relu,
                 weight_decay=0.0001, lr=0.05, epoch=100, device='cuda:0', batch_size=0, num_neigh=-1,
                 verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        self.gnn = gnn

    def init_model(self, **kwargs):
        return GNNBase(in_dim=self.in_dim, hid_dim=self.hid_dim, num_classes=self.num_classes,
                       num_layers=self.num_layers, dropout=self.dropout, gnn=self.gnn, **kwargs).to(self.device)

    def forward_model(self, source_data, target_data):
        source_logits = self.gnn(source_data.x, source_data.edge_index)
        target_logits = self.gnn(target_data.x, target_data.edge_index)
        # GNNBase already returns log-probabilities; the reference applies log_softmax again (:146)
        return source_ce(source_logits, source_data.y), source_logits, target_logits

    def fit(self, source_data, target_data):
        import time
        full = self.batch_size == 0
        sb = source_data.x.shape[0] if full else self.batch_size
        tb = target_data.x.shape[0] if full else self.batch_size
        kw = {} if full else dict(device=self.device)
        source_loader = NeighborLoader(source_data, self.num_neigh, batch_size=sb, **kw)
        target_loader = NeighborLoader(target_data, self.num_neigh, batch_size=tb, **kw)
        self.gnn = self.init_model(**self.kwargs)       # note: replaces the backbone name, as in the reference
        optimizer = torch.optim.Adam(self.gnn.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        start = time.time()
        for epoch in range(self.epoch):
            epoch_loss, logits, labels = 0.0, [], []
            for src, tgt in zip(source_loader, target_loader):
                self.gnn.train()
                src, tgt = src.to(self.device), tgt.to(self.device)
                loss, _, _ = self.forward_model(src, tgt)
                optimizer.zero_grad()
                loss.backward()
                _allreduce_grads(optimizer)
                optimizer.step()
                epoch_loss += loss.item()
                lg, lb = self.predict(src)                                        # :191-195
                logits.append(lg); labels.append(lb)
            acc = eval_micro_f1(torch.cat(labels), torch.cat(logits).argmax(dim=1))
            secs = time.time() - start
            logger(epoch=epoch, loss=epoch_loss, source_train_acc=acc, time=secs, verbose=self.verbose, train=True)
            if self.epoch_hook is not None:
                self.epoch_hook(epoch, epoch_loss, acc, secs)

    def process_graph(self, data):
        pass

    def predict(self, data):
        self.gnn.eval()
        data = data.to(self.device)
        with torch.no_grad():
            logits = self.gnn(data.x, data.edge_index)
        return logits, data.y
