"""``SpecReg`` trainer (pygda/models/specreg.py:18-419): UDAGCN's dual-view encoder trained with a
Wasserstein critic (5 critic updates per encoder update, gradient penalty at the encodings) and
spectral regularisers -- smoothness and maximum-frequency-response hinges on the encodings
projected onto the graph-Laplacian basis ``data.eivec`` ([k, N], pygda/utils/svd_transform.py).

The encoder runs on the MI355X aggregation kernels; the projection ``eivec @ encoded`` is a plain
library GEMM on the matrix cores; the critic is a 3-layer MLP whose double backward stays in
torch autograd.  The critic loop already sees detached encodings in the reference (:187)."""
import itertools
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..metrics import eval_micro_f1
from ..nn import UDAGCNBase
from ..utils import logger
from .base import BaseGDA, _allreduce_grads


class SpecReg(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=3, dropout=0., act=F.relu, ppmi=True,
                 adv_dim=40, reg_mode=True, gamma_adv=0.1, thr_smooth=-1, gamma_smooth=0.01, thr_mfr=-1,
                 gamma_mfr=0.01, weight_decay=3e-3, lr=4e-3, epoch=100, device='cuda:0', batch_size=0,
                 num_neigh=-1, verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        self.ppmi, self.adv_dim, self.reg_mode, self.gamma_adv = ppmi, adv_dim, reg_mode, gamma_adv
        self.thr_smooth, self.gamma_smooth, self.thr_mfr, self.gamma_mfr = thr_smooth, gamma_smooth, thr_mfr, gamma_mfr

    def init_model(self, **kwargs):
        return UDAGCNBase(in_dim=self.in_dim, hid_dim=self.hid_dim, num_classes=self.num_classes,
                          num_layers=self.num_layers, dropout=self.dropout, act=self.act, ppmi=self.ppmi,
                          adv_dim=self.adv_dim, **kwargs).to(self.device)

    def forward_model(self, source_data, target_data, alpha, epoch):
        net = self.udagcn
        encoded_source = net.encode(source_data, "source")
        encoded_target = net.encode(target_data, "target")
        source_logits = net.cls_model(encoded_source)
        cls_loss = net.loss_func(source_logits, source_data.y)                           # :185
        _x_src, _x_tgt = encoded_source.detach(), encoded_target.detach()
        for _ in range(5):                                                               # :188-194
            self.optimizer_critic.zero_grad()
            loss_1 = self.critic(_x_src).mean() - self.critic(_x_tgt).mean()
            loss_2 = self.calculate_gradient_penalty(_x_src, _x_tgt)
            (-loss_1 + 10 * loss_2).backward()
            _allreduce_grads(self.optimizer_critic)      # data-parallel replicas: one critic
            self.optimizer_critic.step()
        loss_grl = self.critic(encoded_source).mean() - self.critic(encoded_target).mean()
        loss = cls_loss + loss_grl * self.gamma_adv
        if self.reg_mode:                                                                # :199-209
            x_src = source_data.eivec @ encoded_source
            x_tgt = target_data.eivec @ encoded_target
            if self.thr_smooth > 0:
                delta_src = (x_src[:-1] - x_src[1:]).abs()
                delta_tgt = (x_tgt[:-1] - x_tgt[1:]).abs()
                loss = loss + (F.relu(delta_src - self.thr_smooth).mean()
                               + F.relu(delta_tgt - self.thr_smooth).mean()) * self.gamma_smooth
            if self.thr_mfr > 0:
                loss = loss + (F.relu(x_src.abs() - self.thr_mfr).mean()
                               + F.relu(x_tgt.abs() - self.thr_mfr).mean()) * self.gamma_mfr
        target_logits = net.cls_model(encoded_target)
        target_probs = torch.clamp(F.softmax(target_logits, dim=-1), min=1e-9, max=1.0)
        loss_entropy = torch.mean(torch.sum(-target_probs * torch.log(target_probs), dim=-1))
        return loss + loss_entropy * (epoch / self.epoch * 0.01), source_logits, target_logits

    def fit(self, source_data, target_data):
        """specreg.py:228-326.  The per-epoch accuracy is the reference's: source logits re-predicted
        in eval mode AFTER the optimiser step (:307-312), not the training-mode logits."""
        if self.reg_mode and (getattr(source_data, "eivec", None) is None or getattr(target_data, "eivec", None) is None):
            raise AttributeError("SpecReg(reg_mode=True) needs data.eivec on both domains "
                                 "(pygda_amd.utils.svd_transform attaches it)")
        self._node_loaders(source_data, target_data)
        self.udagcn = self.init_model(**self.kwargs)
        params = itertools.chain(*[m.parameters() for m in self.udagcn.models])
        # the shared conv Parameters are listed twice (encoder + ppmi_encoder), as in the reference; its CPU
        # path then updates them twice per step, one after the other.  torch's multi-tensor CUDA kernels would
        # process the two list entries concurrently (one racy update): the per-tensor loop keeps the CPU
        # path's semantics.
        optimizer = torch.optim.Adam(params, lr=self.lr, weight_decay=self.weight_decay, foreach=False)
        self.critic = nn.Sequential(nn.Linear(self.hid_dim, self.hid_dim), nn.ReLU(),
                                    nn.Linear(self.hid_dim, self.hid_dim), nn.ReLU(),
                                    nn.Linear(self.hid_dim, 1)).to(self.device)          # :281-287
        self.optimizer_critic = torch.optim.Adam(self.critic.parameters(), self.lr)
        from ..distributed import broadcast_parameters
        broadcast_parameters(self.udagcn)                # data-parallel replicas start from rank 0's weights
        broadcast_parameters(self.critic)
        start_time = time.time()
        for epoch in range(self.epoch):
            epoch_loss, logits, labels = 0.0, [], []
            alpha = min((epoch + 1) / self.epoch, 0.05)
            for src, tgt in zip(self.source_loader, self.target_loader):
                for m in self.udagcn.models:
                    m.train()
                src, tgt = src.to(self.device), tgt.to(self.device)
                if getattr(src, "n_id", None) is not None:
                    raise NotImplementedError(
                        "SpecReg with sampled mini-batches: the adjacency cache keys ('source'/'target') and "
                        "data.eivec both refer to the whole graph (specreg.py:180-181,200-201)")
                loss, _, _ = self.forward_model(src, tgt, alpha, epoch)
                epoch_loss += loss.item()
                optimizer.zero_grad()
                loss.backward()
                _allreduce_grads(optimizer)
                optimizer.step()
                lg, lb = self.predict(src, source=True)
                logits.append(lg)
                labels.append(lb)
            acc = eval_micro_f1(torch.cat(labels), torch.cat(logits).argmax(dim=1))
            secs = time.time() - start_time
            logger(epoch=epoch, loss=epoch_loss, source_train_acc=acc, time=secs, verbose=self.verbose, train=True)
            if self.epoch_hook is not None:
                self.epoch_hook(epoch, epoch_loss, acc, secs)

    def process_graph(self, data):
        pass

    def predict(self, data, source=False):
        """Unlike the other trainers the reference encodes the ``data`` it is given (:367-378),
        with the adjacency cached under 'source' / 'target' during training."""
        for m in self.udagcn.models:
            m.eval()
        data = data.to(self.device)
        with torch.no_grad():
            encoded = self.udagcn.encode(data, 'source' if source else 'target')
            logits = self.udagcn.cls_model(encoded)
        return logits, data.y

    def calculate_gradient_penalty(self, x_src, x_tgt):
        """:380-419: critic input-gradient norm at the encodings themselves (no interpolation)."""
        x = torch.cat([x_src, x_tgt], dim=0).requires_grad_(True)
        x_out = self.critic(x)
        grad = torch.autograd.grad(outputs=x_out, inputs=x, grad_outputs=torch.ones_like(x_out),
                                   create_graph=True, retain_graph=True, only_inputs=True)[0]
        grad = grad.view(grad.shape[0], -1)
        return torch.mean((grad.norm(2, dim=1) - 1) ** 2)
