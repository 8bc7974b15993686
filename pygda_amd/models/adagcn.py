"""``AdaGCN`` trainer (pygda/models/adagcn.py:18-454): Wasserstein critic with gradient
penalty (10 critic updates per encoder update), then source CE + domain_weight * |E D(s) -
E D(t)|.  The encoder runs on the MI355X aggregation kernels; the critic is a 3-layer MLP
whose double backward (gradient penalty) stays in torch autograd.

The critic update (loss, gradient penalty and all four parameter gradients) runs as the fused closed-form
kernels of csrc/gda_critic.hip; the composed torch-autograd form is kept for critics of another shape.

Savings over the reference with identical results: the encoder's first conv is evaluated once per domain
and step (it is deterministic) and shared by the 11 encoder passes; inside the critic loop the encoder
outputs are detached.  The reference back-propagates the critic loss into the encoder ten
times per step and then discards those gradients (``optimizer.zero_grad()`` at :292 precedes
the only encoder step), i.e. 10 x L wasted backward aggregations per domain."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nn import AdaGCNBase
from .base import BaseGDA, _allreduce_grads


class AdaGCN(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, mode='node', num_layers=3, dropout=0., act=F.relu,
                 gnn_type='gcn', adv_dim=40, gp_weight=5, domain_weight=1, weight_decay=0., lr=4e-3,
                 epoch=100, device='cuda:0', batch_size=0, num_neigh=-1, verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        self.gnn_type, self.adv_dim, self.gp_weight = gnn_type, adv_dim, gp_weight
        self.domain_weight, self.mode = domain_weight, mode
        self.critic_steps = 10                                                        # :169
        import os
        self.use_fused_critic = os.environ.get("PYGDA_AMD_FUSED_CRITIC", "1") == "1"

    def init_model(self, **kwargs):
        return AdaGCNBase(in_dim=self.in_dim, hid_dim=self.hid_dim, num_classes=self.num_classes,
                          num_layers=self.num_layers, dropout=self.dropout, act=self.act,
                          gnn_type=self.gnn_type, mode=self.mode, **kwargs).to(self.device)

    def _critic_gap(self, es, et):
        # data-parallel: E D(s), E D(t) are means over every rank's rows (node-count weighted)
        d = self.discriminator
        if torch.is_grad_enabled() and self._fused_critic(es) and d[0].in_features <= 128:
            # the encoder's loss (:190-193): both means and their input gradients from the two-layer discriminator's row
            # kernels with a sigmoid-mean head -- ~40 library launches each way as two
            from ..ops import critic_means, critic_means_ok
            if critic_means_ok(es, d[0].weight, d[3].weight):
                ms, mt = critic_means(es, et, d[0].weight, d[0].bias, d[3].weight, d[3].bias, d[2].p if d.training else 0.0)
                return self._gmean(ms, es.size(0)) - self._gmean(mt, et.size(0))
        return self._gmean(torch.mean(self.discriminator(es).reshape(-1)), es.size(0)) \
            - self._gmean(torch.mean(self.discriminator(et).reshape(-1)), et.size(0))

    def _encoder_loss(self, cls_loss, es, et):
        """``cls_loss + domain_weight * |E D(e_s) - E D(e_t)|`` (:186-196)."""
        d = self.discriminator
        from ..distributed import active
        if (torch.is_grad_enabled() and not active() and self._fused_critic(es) and d[0].in_features <= 128
                and cls_loss.dim() == 0 and os.environ.get("PYGDA_AMD_FUSED_GAP_LOSS", "1") == "1"):
            from ..ops import critic_abs_gap_loss, critic_means_ok
            if critic_means_ok(es, d[0].weight, d[3].weight):
                return critic_abs_gap_loss(cls_loss, es, et, d[0].weight, d[0].bias, d[3].weight, d[3].bias,
                                           d[2].p if d.training else 0.0, self.domain_weight)
        return cls_loss + torch.abs(self._critic_gap(es, et)) * self.domain_weight

    def _fused_critic(self, es):
        """The closed-form critic update (csrc/gda_critic.hip) applies: the reference's critic
        (Linear -> ReLU -> Dropout -> Linear(., 1) -> Sigmoid) on the GPU, means taken over this process's rows."""
        d = self.discriminator
        from ..distributed import active
        if active() and not getattr(getattr(self, "source_loader", None), "full_batch", False):
            return False                      # node-count weighted global means: the composed path
        return (es.is_cuda and isinstance(d, nn.Sequential) and len(d) == 5 and isinstance(d[0], nn.Linear)
                and isinstance(d[1], nn.ReLU) and isinstance(d[2], nn.Dropout) and isinstance(d[3], nn.Linear)
                and isinstance(d[4], nn.Sigmoid) and d[3].out_features == 1 and d[0].in_features % 4 == 0
                and d[0].in_features <= 256 and d[0].out_features <= 64 and self.use_fused_critic)

    def _interp_indices(self, num_s, num_t, dev):
        """Row pairs of the interpolates (:423-434): the smaller domain twice against the head and the tail
        of the larger one."""
        key = (num_s, num_t, str(dev))
        if not hasattr(self, "_interp_cache"):
            self._interp_cache = {}
        hit = self._interp_cache.get(key)
        if hit is None:
            m = min(num_s, num_t)
            if num_s == num_t:
                i_s = i_t = torch.arange(m)
            elif num_s < num_t:
                i_s = torch.cat([torch.arange(m), torch.arange(m)])
                i_t = torch.cat([torch.arange(m), torch.arange(num_t - m, num_t)])
            else:
                i_s = torch.cat([torch.arange(m), torch.arange(num_s - m, num_s)])
                i_t = torch.cat([torch.arange(m), torch.arange(m)])
            hit = (i_s.to(torch.int32).to(dev), i_t.to(torch.int32).to(dev))
            self._interp_cache[key] = hit
        return hit

    def _batched_encodes(self, h0, source_data, target_data):
        return (self.mode == 'node' and h0.is_cuda and self.critic_steps > 1 and self.adagcn.encoder.gnn_type == 'gcn'
                and all(getattr(d.edge_index, "_gda_static", False) for d in (source_data, target_data))
                and os.environ.get("PYGDA_AMD_BATCHED_CRITIC_ENCODES", "1") == "1")

    def _encode_copies(self, h0, data):
        """``[critic_steps, n, hid]``: the encoder stack after its first conv on `critic_steps` stacked copies."""
        from ..data import Data
        S, n = self.critic_steps, h0.size(0)
        cache = self.__dict__.setdefault("_copies_cache", {})
        key = (data.edge_index.data_ptr(), S, n)
        rep = cache.get(key)
        if rep is None:
            ei = data.edge_index
            off = (torch.arange(S, device=ei.device) * n).repeat_interleave(ei.size(1))
            rep = cache[key] = (ei, Data(x=None, edge_index=ei.repeat(1, S) + off, y=None))   # the entry keeps `ei` alive
            rep[1].edge_index._gda_static = True
        return self.adagcn.forward_from(h0, rep[1], copies=S).view(S, n, -1)

    def _critic_update_fused(self, es, et):
        from ..hipgraph import host_rand
        from ..ops import wgan_critic_adam, wgan_critic_grads
        d = self.discriminator
        idx_s, idx_t = self._interp_indices(es.size(0), et.size(0), es.device)
        alpha = host_rand((idx_s.numel(), 1), es.device)                            # the reference's CPU draw
        params = (d[0].weight, d[0].bias, d[3].weight, d[3].bias)
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if getattr(self, "_critic_loss", None) is None or self._critic_loss.device != es.device:
            self._critic_loss = torch.zeros(1, dtype=torch.float32, device=es.device)
        p_drop = d[2].p if d.training else 0.0
        from ..distributed import active
        if (not active() and os.environ.get("PYGDA_AMD_CRITIC_FUSED_ADAM", "1") == "1"
                and wgan_critic_adam(es, et, idx_s, idx_t, alpha, params, self.c_optimizer, p_drop, self.gp_weight,
                                     self._critic_loss)):
            return                               # gradients AND the optimiser's step in the update's two launches
        wgan_critic_grads(es, et, idx_s, idx_t, alpha, *params, p_drop, self.gp_weight,
                          (self._critic_loss, *(p.grad for p in params)))
        _allreduce_grads(self.c_optimizer)       # data-parallel: replica critics stay identical
        self.c_optimizer.step()

    def forward_model(self, source_data, target_data):
        net = self.adagcn
        both = self._stacked_pair(source_data, target_data) \
            if (torch.is_grad_enabled() and net.encoder.gnn_type == 'gcn' and getattr(source_data, "n_id", None) is None) else None
        if both is not None:
            return self._forward_model_stacked(both, source_data, target_data)
        # the first conv of the encoder (projection + aggregation, nothing random) once per domain and step
        h0_s, h0_t = net.first_conv(source_data), net.first_conv(target_data)
        # The critic loop re-encodes both domains every step (:170-171) with an encoder that does not change inside
        # the loop: its `critic_steps` passes differ by their dropout draws only, so they run as ONE pass over
        # `critic_steps` stacked copies (block-diagonal graph, independent draws per copy): 5 launches per domain
        # instead of 4 per critic step.
        batched = self._batched_encodes(h0_s, source_data, target_data)
        if batched:
            with torch.no_grad():
                es_all = self._encode_copies(h0_s.detach(), source_data)
                et_all = self._encode_copies(h0_t.detach(), target_data)
        for k in range(self.critic_steps):                                            # :169-183
            if batched:
                encoded_source, encoded_target = es_all[k], et_all[k]
            else:
                with torch.no_grad():
                    encoded_source = net.forward_from(h0_s.detach(), source_data)
                    encoded_target = net.forward_from(h0_t.detach(), target_data)
            if self._fused_critic(encoded_source):
                self._critic_update_fused(encoded_source, encoded_target)
                continue
            gp_loss = self.gradient_penalty(encoded_source, encoded_target)
            loss = -torch.abs(self._critic_gap(encoded_source, encoded_target)) + self.gp_weight * gp_loss
            self.c_optimizer.zero_grad()
            loss.backward()
            _allreduce_grads(self.c_optimizer)       # data-parallel: replica critics stay identical
            self.c_optimizer.step()
        encoded_source = net.forward_from(h0_s, source_data)                          # :186-196
        encoded_target = net.forward_from(h0_t, target_data)
        source_logits = self.adagcn.cls_model(encoded_source)
        cls_loss = self._gmean(self._source_loss(source_logits, source_data.y), source_logits.size(0))
        loss = self._encoder_loss(cls_loss, encoded_source, encoded_target)
        target_logits = self.adagcn.cls_model(encoded_target)
        return loss, source_logits, target_logits

    def _source_loss(self, logits, labels):
        """``loss_func(source_logits, y)`` (adagcn.py:189) -- the fused loss kernels for the CrossEntropyLoss the trainer builds."""
        lf = self.adagcn.loss_func
        if isinstance(lf, nn.CrossEntropyLoss):
            from ..ops import source_ce
            return source_ce(logits, labels)
        return lf(logits, labels)

    def _forward_model_stacked(self, both, source_data, target_data):
        """forward_model() with every encoder pass over BOTH domains at once (the block-diagonal pair of
        BaseGDA._stacked_pair): the first conv, the critic loop's `critic_steps` re-encodings (stacked copies of the pair)
        and the encoder update's pass -- half the encoder launches each way; the critic updates are the same calls on the
        same rows."""
        from ..ops import split_rows
        net, ns = self.adagcn, both.ns
        h0 = net.first_conv(both)
        batched = self._batched_encodes(h0, source_data, target_data)
        if batched:
            with torch.no_grad():
                e_all = self._encode_copies(h0.detach(), both)            # [critic_steps, ns + nt, hid]
        for k in range(self.critic_steps):                                            # :169-183
            if batched:
                encoded_source, encoded_target = e_all[k, :ns], e_all[k, ns:]
            else:
                with torch.no_grad():
                    e = net.forward_from(h0.detach(), both)
                encoded_source, encoded_target = e[:ns], e[ns:]
            if self._fused_critic(encoded_source):
                self._critic_update_fused(encoded_source, encoded_target)
                continue
            gp_loss = self.gradient_penalty(encoded_source, encoded_target)
            loss = -torch.abs(self._critic_gap(encoded_source, encoded_target)) + self.gp_weight * gp_loss
            self.c_optimizer.zero_grad()
            loss.backward()
            _allreduce_grads(self.c_optimizer)
            self.c_optimizer.step()
        encoded_source, encoded_target = split_rows(net.forward_from(h0, both), ns)  # :186-196
        source_logits = net.cls_model(encoded_source)
        cls_loss = self._gmean(self._source_loss(source_logits, source_data.y), source_logits.size(0))
        loss = self._encoder_loss(cls_loss, encoded_source, encoded_target)
        target_logits = net.cls_model(encoded_target)
        return loss, source_logits, target_logits

    def _prepare(self, source_data, target_data):
        # mode='graph' (adagcn.py:244-252, adagcn_base.py:93-94): shuffled DataLoader batches, the encoder's output
        # mean-pooled per graph -- critic, gradient penalty and classifier then see one row per graph
        self._loaders(source_data, target_data)
        self.adagcn = self.init_model(**self.kwargs)
        on_gpu = torch.device(self.device).type == "cuda"
        if on_gpu:       # torch.optim.Adam's rule in one capturable launch (pygda_amd/optim.py)
            from ..optim import Adam
        else:
            Adam = torch.optim.Adam
        optimizer = Adam(self.adagcn.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        self.discriminator = nn.Sequential(nn.Linear(self.hid_dim, self.adv_dim), nn.ReLU(), nn.Dropout(0.1),
                                           nn.Linear(self.adv_dim, 1), nn.Sigmoid()).to(self.device)   # :264-270
        self.c_optimizer = Adam(self.discriminator.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        # no per-epoch scalar enters the step, its host draws go through hipgraph.host_rand and the critic's
        # optimiser is rolled back with the encoder's: the step (10 critic updates + encoder update) replays
        self._graph_safe_step = self.mode == 'node'      # graph mode re-collates a shuffled batch every epoch
        self._graph_extra_optimizers = [self.c_optimizer]
        self._dp_aux_modules = [self.discriminator]      # broadcast from rank 0 with the encoder

        def step(src, tgt, alpha, epoch):
            loss, source_logits, _ = self.forward_model(src, tgt)
            return loss, source_logits

        return self.adagcn, optimizer, step, lambda e: 0.0

    def fit(self, source_data, target_data):
        self._train_epochs(*self._prepare(source_data, target_data))

    def process_graph(self, data):
        pass

    def predict(self, data, source=False):
        self.adagcn.eval()
        loader = self.source_loader if source else self.target_loader
        return self._predict_loader(loader, lambda b: self.adagcn.cls_model(self.adagcn(b)))

    def gradient_penalty(self, encoded_source, encoded_target):
        """WGAN-GP over cat(source, target, interpolates) (:387-454); interpolation weights from
        the CPU generator, as in the reference (``torch.rand(...).to(device)``) -- through
        ``hipgraph.host_rand`` so that a captured step is fed the same draws."""
        from ..hipgraph import host_rand
        num_s, num_t = encoded_source.shape[0], encoded_target.shape[0]
        dev = encoded_source.device
        if num_s < num_t:
            hidden_s = torch.cat((encoded_source, encoded_source), dim=0)
            hidden_t = torch.cat((encoded_target[0:num_s], encoded_target[-num_s:]), dim=0)
            alpha = host_rand((2 * num_s, 1), dev)
        elif num_s > num_t:
            hidden_s = torch.cat((encoded_source[0:num_t], encoded_source[-num_t:]), dim=0)
            hidden_t = torch.cat((encoded_target, encoded_target), dim=0)
            alpha = host_rand((2 * num_t, 1), dev)
        else:
            hidden_s, hidden_t = encoded_source, encoded_target
            alpha = host_rand((num_t, 1), dev)
        interpolates = hidden_t + alpha * (hidden_s - hidden_t)
        inputs = torch.cat((encoded_source, encoded_target, interpolates), dim=0)
        if not inputs.requires_grad:
            inputs.requires_grad_(True)
        scores = self.discriminator(inputs)
        gradient = torch.autograd.grad(inputs=inputs, outputs=scores, grad_outputs=torch.ones_like(scores),
                                       create_graph=True, retain_graph=True, only_inputs=True)[0]
        gradient_norm = gradient.view(gradient.shape[0], -1).norm(2, dim=1)
        return self._gmean(torch.mean((gradient_norm - 1) ** 2), gradient_norm.size(0))
