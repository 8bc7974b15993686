"""``DGSDA`` trainer (pygda/models/dgsda.py:17-436): source CE + alpha * L1(theta_s, theta_t) +
beta * MMD(relu(lin1 x_s), relu(lin1 x_t)) + gamma * frequency-weighted target entropy, over the
Bernstein-filter network of :mod:`pygda_amd.nn.dgsda_base`."""
import torch
import torch.nn.functional as F

from ..ops import source_ce
from ..nn.dgsda_base import DGSDABase
from ..utils import MMD
from .base import BaseGDA


class DGSDA(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, mode='node', num_layers=2, dropout=0., act=F.relu, K=8,
                 alpha=0.05, beta=0.5, gamma=0.05, weight_decay=0., lr=4e-3, epoch=200, device='cuda:0',
                 batch_size=0, num_neigh=-1, verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        assert num_layers == 2, 'unsupport number of layers'                              # dgsda.py:110-111
        assert mode == 'node', 'unsupport mode'
        self.K, self.mode, self.alpha, self.beta, self.gamma = K, mode, alpha, beta, gamma

    def init_model(self, **kwargs):
        return DGSDABase(features=self.in_dim, hidden=self.hid_dim, classes=self.num_classes,
                         dprate=self.dropout, K=self.K, **kwargs).to(self.device)

    def forward_model(self, source_data, target_data):
        net = self.dgsda
        source_logits = net(source_data)                                                  # :172
        loss = source_ce(source_logits, source_data.y)
        loss = loss + F.l1_loss(net.prop1.temp, net.prop2.temp) * self.alpha              # :176-179
        source_feature = F.relu(net.lin1(source_data.x))
        target_feature = F.relu(net.lin1(target_data.x))
        loss = loss + MMD(source_feature, target_feature) * self.beta                     # :181-185
        target_outputs = net(target_data, False)
        loss = loss + self.entropy_minimization_loss(target_outputs) * self.gamma         # :187-190
        return loss, source_logits

    def entropy_minimization_loss(self, output):                                          # :198-227
        probs = F.softmax(output, dim=1)
        log_probs = F.log_softmax(output, dim=1)
        a = torch.sum(probs, dim=0)
        return -torch.sum(probs * log_probs / (a / torch.sum(a)), dim=1).mean()

    def fit(self, source_data, target_data):
        self._node_loaders(source_data, target_data)
        self.dgsda = self.init_model(**self.kwargs)
        self._graph_safe_step = True          # nothing in the step depends on per-epoch Python scalars
        on_gpu = torch.device(self.device).type == "cuda"
        if on_gpu:
            from ..optim import Adam
            optimizer = Adam(self.dgsda.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        else:
            optimizer = torch.optim.Adam(self.dgsda.parameters(), lr=self.lr, weight_decay=self.weight_decay)

        def step(src, tgt, alpha, epoch):
            return self.forward_model(src, tgt)

        self._train_epochs(self.dgsda, optimizer, step, lambda e: 0.0)

    def process_graph(self, data):
        pass

    def predict(self, data, source=False):
        self.dgsda.eval()
        loader = self.source_loader if source else self.target_loader
        return self._predict_loader(loader, lambda b: self.dgsda(b, source))
