"""``UDAGCN`` trainer (pygda/models/udagcn.py:19-360): source CE + gradient-reversed domain
CE per domain (MLP discriminator) + annealed target entropy, over the dual-view encoder.

The reference's per-name adjacency cache is never invalidated (cached_gcn_conv.py:132-136),
so with ``batch_size > 0`` it would silently reuse batch #1's edges for every later batch.
Here mini-batches key the cache per batch (SURVEY §3.4 hazard); full-batch runs use the
reference's keys "source"/"target" unchanged.

``mode='graph'`` (udagcn.py:168-170, 248-256, 360-377): node embeddings are mean-pooled per graph before the
classifier, the discriminator and the entropy term.  Its loaders shuffle (``DataLoader(..., shuffle=True)``), so
in the reference EVERY batch after the first meets the first batch's cached adjacency -- other graphs' edges
applied to this batch's nodes, or an index error when the node counts differ.  Graph-mode batches are keyed per
batch here as well (the goldens are recorded from the reference with its caches emptied before every step)."""
import itertools

import torch
import torch.nn.functional as F

from ..nn import GradReverse, UDAGCNBase
from ..nn.a2gnn_base import global_mean_pool
from .base import BaseGDA


import os

FUSED_DOMAIN_MODEL = os.environ.get("PYGDA_AMD_FUSED_DOMAIN_MODEL", "1") == "1"


class UDAGCN(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, mode='node', num_layers=2, dropout=0., act=F.relu,
                 ppmi=True, adv_dim=40, weight_decay=3e-3, lr=4e-3, epoch=300, device='cuda:0',
                 batch_size=0, num_neigh=-1, verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        self.ppmi, self.adv_dim, self.mode = ppmi, adv_dim, mode

    def init_model(self, **kwargs):
        return UDAGCNBase(in_dim=self.in_dim, hid_dim=self.hid_dim, num_classes=self.num_classes,
                          num_layers=self.num_layers, dropout=self.dropout, act=self.act, ppmi=self.ppmi,
                          adv_dim=self.adv_dim, **kwargs).to(self.device)

    def _cache_key(self, data, name):
        """Full batch: the reference's keys.  Sampled mini-batch: ONE transient key per domain whose
        entries are dropped from every conv layer before the batch is encoded -- the graphs of a batch
        live for that batch only (no growth with the number of steps, no host sync for a key, no stale
        hit when two batches happen to agree in their first seed and sizes)."""
        if getattr(data, "n_id", None) is None and self.mode == 'node':
            return name
        key = name + ":minibatch"
        for enc in (self.udagcn.encoder, getattr(self.udagcn, "ppmi_encoder", None)):
            for conv in (() if enc is None else enc.conv_layers):
                conv.cache_dict.pop(key, None)
        return key

    def _domain_loss(self, net, encoded_source, encoded_target, alpha, ns, nt):
        """``CE(domain_model(GRL(source)), 0) + CE(domain_model(GRL(target)), 1)`` (udagcn.py:176-190): on the GPU one
        fused row kernel and one fold each way (ops.grl_mlp_ce) for the discriminator the reference builds
        (Linear - ReLU - Dropout - Linear(., 2), udagcn_base.py:157-162); anything else, and the CPU, composes it."""
        dm, gm = net.domain_model, self._gmean
        from ..distributed import active
        from ..ops import grl_mlp_ce, grl_mlp_ce_ok
        if (FUSED_DOMAIN_MODEL and len(dm) == 4 and isinstance(dm[0], torch.nn.Linear) and isinstance(dm[1], torch.nn.ReLU)
                and isinstance(dm[2], torch.nn.Dropout) and isinstance(dm[3], torch.nn.Linear)
                and grl_mlp_ce_ok(encoded_source, dm[0].weight, dm[3].weight)):
            p = dm[2].p if dm.training else 0.0
            args = (encoded_source, encoded_target, dm[0].weight, dm[0].bias, dm[3].weight, dm[3].bias, alpha, p)
            if active() and not getattr(self.source_loader, "full_batch", False):     # node-count weighted means
                ls, lt = grl_mlp_ce(*args, pair=True)
                return gm(ls, ns) + gm(lt, nt)
            return grl_mlp_ce(*args)
        dev = encoded_source.device
        source_domain_preds = dm(GradReverse.apply(encoded_source, alpha))
        target_domain_preds = dm(GradReverse.apply(encoded_target, alpha))
        return gm(net.loss_func(source_domain_preds,
                                torch.zeros(source_domain_preds.size(0), dtype=torch.long, device=dev)), ns) \
            + gm(net.loss_func(target_domain_preds,
                               torch.ones(target_domain_preds.size(0), dtype=torch.long, device=dev)), nt)

    def _stacked_caches(self):
        """Every conv layer's cached operator of the block-diagonal (source, target) pair, combined from the two it cached
        for the domains (the PPMI graphs are NOT rebuilt: their walks were drawn per domain, in the reference's order);
        False until both exist -- the first step therefore runs the two passes."""
        from ..graph import block_diag
        for enc in (self.udagcn.encoder, getattr(self.udagcn, "ppmi_encoder", None)):
            for conv in (() if enc is None else enc.conv_layers):
                if "source+target" not in conv.cache_dict:
                    gs, gt = conv.cache_dict.get("source"), conv.cache_dict.get("target")
                    if gs is None or gt is None:
                        return False
                    conv.cache_dict["source+target"] = block_diag(gs, gt)
        return True

    def forward_model(self, source_data, target_data, alpha, epoch):
        net = self.udagcn
        both = self._stacked_pair(source_data, target_data) \
            if (torch.is_grad_enabled() and getattr(source_data, "n_id", None) is None) else None
        if both is not None and self._stacked_caches():
            # full-batch node mode: the encoder over both domains as ONE pass over the block-diagonal pair (:165-166)
            from ..ops import split_rows
            encoded_source, encoded_target = split_rows(net.encode(both, "source+target"), both.ns)
        else:
            encoded_source = net.encode(source_data, self._cache_key(source_data, "source"))
            encoded_target = net.encode(target_data, self._cache_key(target_data, "target"))
        if self.mode == 'graph':                                                              # :168-170
            encoded_source = global_mean_pool(encoded_source, source_data.batch)
            encoded_target = global_mean_pool(encoded_target, target_data.batch)
        source_logits = net.cls_model(encoded_source)
        gm = self._gmean
        ns, nt = encoded_source.size(0), encoded_target.size(0)
        from ..ops import softmax_entropy, source_ce
        ce = source_ce(source_logits, source_data.y) if isinstance(net.loss_func, torch.nn.CrossEntropyLoss) \
            else net.loss_func(source_logits, source_data.y)          # (the fused loss kernels; same mean over the rows)
        loss = gm(ce, ns)                                                                    # :172
        dev = encoded_source.device
        loss = loss + self._domain_loss(net, encoded_source, encoded_target, alpha, ns, nt)   # :176-190
        target_logits = net.cls_model(encoded_target)
        loss_entropy = gm(softmax_entropy(target_logits, 1e-9), nt)                            # :193-197
        return loss + loss_entropy * (epoch / self.epoch * 0.01), source_logits, target_logits

    def _prepare(self, source_data, target_data):
        self._loaders(source_data, target_data)                                              # :224-258
        self.udagcn = self.init_model(**self.kwargs)
        params = itertools.chain(*[m.parameters() for m in self.udagcn.models])            # :262-268
        # the shared conv Parameters are listed twice (encoder + ppmi_encoder), as in the reference; its CPU
        # path then updates them twice per step, one after the other.  torch's multi-tensor CUDA kernels would
        # process the two list entries concurrently (one racy update): the per-tensor loop keeps the CPU
        # path's semantics.
        if torch.device(self.device).type == "cuda":
            # one capturable multi-tensor launch per round of duplicates (pygda_amd/optim.py): the second listing
            # of a shared Parameter is updated after the first, as in the per-tensor loop
            from ..optim import Adam
            optimizer = Adam(list(params), lr=self.lr, weight_decay=self.weight_decay)
            # the step's per-epoch scalars (GRL alpha, entropy weight) reach the kernels as 0-dim device tensors
            # refreshed before every replay: the full-batch step replays as a hipGraph
            self._graph_safe_step, self._graph_uses_scalars = self.mode == 'node', True
        else:
            optimizer = torch.optim.Adam(params, lr=self.lr, weight_decay=self.weight_decay, foreach=False)

        def step(src, tgt, alpha, epoch):
            loss, source_logits, _ = self.forward_model(src, tgt, alpha, epoch)
            return loss, source_logits

        def before_step():
            for m in self.udagcn.models:
                m.train()

        return self.udagcn, optimizer, step, lambda e: min((e + 1) / self.epoch, 0.05), before_step

    def fit(self, source_data, target_data):
        self._train_epochs(*self._prepare(source_data, target_data))

    def process_graph(self, data):
        pass

    def predict(self, data, source=False):
        for m in self.udagcn.models:
            m.eval()
        loader, name = (self.source_loader, 'source') if source else (self.target_loader, 'target')
        def forward(b):                                                                       # :356-378
            enc = self.udagcn.encode(b, self._cache_key(b, name))
            return self.udagcn.cls_model(global_mean_pool(enc, b.batch) if self.mode == 'graph' else enc)

        return self._predict_loader(loader, forward)
