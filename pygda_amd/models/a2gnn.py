"""``A2GNN`` trainer (pygda/models/a2gnn.py:17-411) on the MI355X kernels: source
cross-entropy + weight * (sampled multi-kernel MMD | gradient-reversed domain CE) with
asymmetric propagation depths ``s_pnums`` / ``t_pnums``."""
import numpy as np
import torch
import torch.nn.functional as F

from ..nn import A2GNNBase
from ..ops import grl_disc_ce, source_ce
from ..utils import MMD
from .base import BaseGDA


import contextlib

_null = contextlib.nullcontext


EARLY_CE_BACKWARD = __import__("os").environ.get("PYGDA_AMD_EARLY_CE_BACKWARD", "1") == "1"


class A2GNN(BaseGDA):
    def __init__(self, in_dim, hid_dim, num_classes, mode='node', num_layers=3, dropout=0.,
                 act=F.relu, s_pnums=0, t_pnums=30, adv=False, weight=5, weight_decay=0., lr=4e-3,
                 epoch=200, device='cuda:0', batch_size=0, num_neigh=-1, verbose=2, **kwargs):
        super().__init__(in_dim=in_dim, hid_dim=hid_dim, num_classes=num_classes, num_layers=num_layers,
                         dropout=dropout, act=act, weight_decay=weight_decay, lr=lr, epoch=epoch,
                         device=device, batch_size=batch_size, num_neigh=num_neigh, verbose=verbose,
                         **kwargs)
        self.s_pnums, self.t_pnums, self.adv, self.weight, self.mode = s_pnums, t_pnums, adv, weight, mode
        # the reference also runs a second full target forward per step whose logits it returns
        # and never uses (a2gnn.py:211); kept by default so a step does the reference's work
        self.compute_target_logits = True
        import os
        self.overlap_streams = os.environ.get("PYGDA_AMD_OVERLAP", "1") == "1"
        self.split_graphs = os.environ.get("PYGDA_AMD_SPLIT_GRAPHS", "0") == "1"   # measured slower (profiles/HISTORY.md 4.7): opt-in
        # sampled mini-batches: the source branch (s_pnums = 0: projections and activations, chip-filling kernels) beside
        # the target branch (K-step launches over ~17 k interior rows: latency-sized) on two streams -- cfg-S, 40 steps,
        # three runs each on one box: 2.88 / 2.98 / 2.92 ms/step against 3.10 / 3.11 / 4.11 on one stream
        # (profiles/r4_stream_experiments.txt); eager launches, so none of the forked-graph scheduling of profiles/HISTORY.md 4.7
        self.overlap_sampled = os.environ.get("PYGDA_AMD_SAMPLED_OVERLAP", "1") == "1"
        self.features_first = os.environ.get("PYGDA_AMD_FEATURES_FIRST", "1") == "1"
        # the step reads nothing but the loaders' batches: a large power-law full-batch graph may be trained on its
        # degree-ordered relabelling (pygda_amd/data.py::auto_reorder; predict() maps the rows back)
        self._auto_reorder_ok = type(self) is A2GNN

    def init_model(self, **kwargs):
        return A2GNNBase(in_dim=self.in_dim, hid_dim=self.hid_dim, num_classes=self.num_classes,
                         num_layers=self.num_layers, adv=self.adv, dropout=self.dropout, act=self.act,
                         mode=self.mode, **kwargs).to(self.device)

    def _target_logits_async(self, net, target_data, h0, after=None):
        """The reference's second target forward (:211) is not part of the loss.  It is issued on
        a side HIP stream (fork/join, no autograd tape) so that its small aggregation launches
        overlap the loss branch instead of queueing behind it; under hipGraph capture the fork
        becomes a parallel branch of the graph.  ``after``: an event recorded where ``h0`` was complete -- the side
        stream then waits for that point instead of for everything issued on this stream since."""
        main = torch.cuda.current_stream()
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream()
        if after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            feats = net.feat_bottleneck_from(h0.detach(), target_data.edge_index, None, self.t_pnums, draw=1)
            out = net.feat_classifier(feats, target_data.edge_index, None, 1)
        out.record_stream(main)
        return out, side

    def _source_branch(self, source_data):
        """Source logits pass (:181-182) and source feature pass (:192) over one shared layer 0."""
        net = self.a2gnn
        sb = None if self.mode == 'node' else source_data.batch
        # sampled batches with s_pnums = 0: layer 0's projection writes both passes' activations itself ("stacked")
        h0_s = net.first_conv(source_data.x, source_data.edge_index, self.s_pnums,
                              draws="stacked" if (self.s_pnums <= 0 and getattr(source_data, "n_id", None) is not None) else 0)
        # The feature pass (:192) is issued BEFORE the logits pass (:181-182) although the reference runs them the
        # other way round (independent passes, same values): autograd walks newer nodes first, so the backward of
        # the logits / cross-entropy path -- which needs nothing from the domain loss -- is enqueued on this
        # branch's stream first and runs beside the MMD's backward kernel instead of queueing behind the feature
        # path that has to wait for it (~100 us off the step's critical path).
        # With s_pnums = 0 (the default) the two passes are one pass over stacked rows (A2GNNBase.feat_pair_from).
        source_features, feats = net.feat_pair_from(h0_s, source_data.edge_index, sb, self.s_pnums)   # :192, :181
        source_logits = net.feat_classifier(feats, source_data.edge_index, sb, 1)            # :181
        loss = self._gmean(source_ce(source_logits, source_data.y), source_logits.size(0))   # :182, fused
        from .. import hipgraph
        self._early = self._src_ready = None
        if (EARLY_CE_BACKWARD and hipgraph.defer_total and type(self) is A2GNN and not self.adv and self.mode == 'node'
                and torch.is_grad_enabled() and feats.requires_grad):
            # Captured step: the part of the backward pass that hangs off the cross-entropy alone -- loss -> logits ->
            # classifier aggregation and its two products, down to the classifier's input `feats` -- is issued HERE, on
            # this branch's stream, right behind the loss kernel, instead of behind the domain loss's forward kernels
            # (one engine run seeds all terms from the stream the MMD was launched on: the 8 kernels of this chain waited
            # for the 55 - 80 us MMD kernel and then led the source branch's backward, the step's last chain to finish).
            # The rest of the backward pass continues from `feats` with this gradient as a second root
            # (GraphedStep._run); the classifier's parameters receive their only gradient here.  Same kernels, same sums.
            # Measured (cfg-A, alternating runs on one box): 0.4177 / 0.4127 / 0.4009 against 0.4239 / 0.4256 / 0.4215.
            # Issuing the target feature pass ahead of the source branch as well, this chain and the loss-unused
            # logits pass behind the domain loss and the backward pass as ordered engine runs was built and measured in
            # the same session: 0.48 - 0.60 ms (the runtime puts the source branch on the logits pass's queue, as in
            # round 2 -- profiles/HISTORY.md 4.7) -- removed again.
            cls_params = [p for p in net.cls.parameters() if p.requires_grad]
            self._src_ready = torch.cuda.Event()      # what the other branches join on: the forward pass, not this chain
            self._src_ready.record()
            grads = torch.autograd.grad(loss, [feats] + cls_params)
            self._early = (feats, grads[0], cls_params, grads[1:])
            loss = loss.detach()
        return loss, source_logits, source_features

    def _target_branch(self, target_data, fork, h0_t=None):
        """Target feature pass (:193); the loss-unused logits pass (:211) forked from its layer 0."""
        net = self.a2gnn
        tb = None if self.mode == 'node' else target_data.batch
        table = getattr(self, "_grad_aliases", None)
        with (net.second_leaves(table) if table is not None and torch.is_grad_enabled() else _null()):
            if h0_t is None:
                # sampled batches: layer 0's activation (both passes' dropout draws) rides in the aggregation's epilogue
                h0_t = net.first_conv(target_data.x, target_data.edge_index, self.t_pnums,
                                      draws=(2 if self.compute_target_logits else 1)
                                      if getattr(target_data, "n_id", None) is not None else 0)
            pending = None
            if self.compute_target_logits and fork and self.features_first:
                # The feature pass (:193) feeds the domain loss, the logits pass (:211) feeds nothing: the pass issued
                # FIRST is the one a replayed graph starts first -- the other branch of the fork sat 15-35 us behind it
                # on every timeline (profiles/r4_cfgA_timeline.txt) -- so the feature pass goes first, as in the
                # reference, and the logits pass forks from an event recorded where layer 0 was complete.
                ready = torch.cuda.Event()
                ready.record()
                target_features = net.feat_bottleneck_from(h0_t, target_data.edge_index, tb, self.t_pnums)   # :193
                pending = self._target_logits_async(net, target_data, h0_t, after=ready)
                return h0_t, pending, target_features
            if self.compute_target_logits and fork:
                pending = self._target_logits_async(net, target_data, h0_t)
            target_features = net.feat_bottleneck_from(h0_t, target_data.edge_index, tb, self.t_pnums)   # :193
        return h0_t, pending, target_features

    def _domain_loss(self, loss, source_features, target_features, alpha):
        net = self.a2gnn
        if self.adv and self.mode != 'node':
            # the reference sizes its domain labels by NODE counts (:199-203) while the pooled features have one row
            # per graph: F.cross_entropy raises ValueError there -- the same error, named
            raise ValueError(f"Expected input batch_size ({source_features.size(0) + target_features.size(0)}) to match "
                             "target batch_size (the node count): adv=True is not defined for mode='graph' in pygda "
                             "(a2gnn.py:199-204)")
        if self.adv:                                                                     # :196-205, fused
            disc = net.domain_discriminator
            dom = grl_disc_ce(source_features, target_features, disc.weight, disc.bias, alpha)
            return loss + self.weight * self._gmean(dom, source_features.size(0) + target_features.size(0))
        # the weight rides inside the loss kernels; the CE term is added OUTSIDE on purpose: as an input of the MMD node
        # its gradient would only be released after the MMD's backward kernels, serialising the CE path behind them
        mmd = MMD(source_features, target_features, scale=self.weight)                  # :206-209
        from .. import hipgraph
        if hipgraph.defer_total and type(self) is A2GNN:
            terms = hipgraph.LossTerms((loss, mmd))      # summed beside the backward pass (hipgraph.LossTerms)
            early = getattr(self, "_early", None)
            if early is not None:
                feats, g_feats, cls_params, g_cls = early
                terms.extra_roots, terms.preset_grads = [(feats, g_feats)], list(zip(cls_params, g_cls))
                self._early = None
            return terms
        return loss + mmd

    def _split_graph_parts(self):
        """The step as three captures (pygda_amd/hipgraph.py::GraphedStepSplit): source forward and target
        forward as separate single-branch graphs replayed on two streams at once, then the domain loss, the
        backward pass and the optimiser step.  A forked capture serialises branches forked at its root on this
        runtime and pays 5-6 us per kernel, so the source branch of the one-graph step runs BEFORE the target
        branch; as two graphs they run side by side."""
        if type(self) is not A2GNN or self.mode != 'node' or not self.overlap_streams or not self.split_graphs:
            return None

        def src_part(src):
            return self._source_branch(src)

        def tgt_part(tgt):
            h0_t, pending, tf = self._target_branch(tgt, True)
            if pending is not None:
                torch.cuda.current_stream().wait_stream(pending[1])
                return tf, pending[0]
            return (tf,)

        def rest_part(src_out, tgt_out, alpha):
            ce, source_logits, sf = src_out
            return self._domain_loss(ce, sf, tgt_out[0], alpha), source_logits

        return src_part, tgt_part, rest_part

    def _branches(self, source_data, target_data):
        """Everything of a step up to the domain loss: ``(ce_loss, source_logits, source_features,
        target_features, h0_t, pending)``; ``pending`` = the forked target logits pass or None."""
        net = self.a2gnn
        node = self.mode == 'node'
        sb = None if node else source_data.batch
        tb = None if node else target_data.batch
        # Full-batch graphs: every kernel of a step is a few microseconds, so a step costs (kernels on
        # the critical path) x (dependent-launch latency), not work.  The source branch, the target
        # feature branch and the loss-unused target logits pass are independent until the domain
        # loss, so they are issued on three HIP streams (fork/join; parallel branches of the hipGraph
        # under capture).  Autograd runs every backward node on its forward stream, so the backward
        # pass splits the same way.  Sampled mini-batches ingest new graphs every step on the main
        # stream and keep the single-stream order.
        sampled = getattr(target_data, "n_id", None) is not None or getattr(source_data, "n_id", None) is not None
        fork = node and source_data.x.is_cuda and self.overlap_streams and (not sampled or self.overlap_sampled)
        main = torch.cuda.current_stream() if fork else None
        if fork:
            src_stream = getattr(self, "_src_stream", None)
            if src_stream is None:
                src_stream = self._src_stream = torch.cuda.Stream()
                # parameters are shared by branches on different streams on purpose; autograd syncs
                # the accumulation itself and only warns about the mismatch
                quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
                if quiet is not None:
                    quiet(False)
            src_stream.wait_stream(main)
        with (torch.cuda.stream(src_stream) if fork else _null()):
            loss, source_logits, source_features = self._source_branch(source_data)
        h0_t, pending, target_features = self._target_branch(target_data, fork)
        if fork:
            ready = getattr(self, "_src_ready", None)
            if ready is not None:                       # (the early cross-entropy backward stays on its branch: the engine
                main.wait_event(ready)                  # run of the remaining backward pass joins that stream at its end)
            else:
                main.wait_stream(src_stream)                                             # join
            for t in (loss, source_logits, source_features):
                t.record_stream(main)
        return loss, source_logits, source_features, target_features, h0_t, pending, (sb, tb)

    def forward_model(self, source_data, target_data, alpha):
        """a2gnn.py:146-213.  Layer 0 (projection + prop_nums aggregations, no randomness) is
        evaluated once per domain and shared by the passes that the reference runs separately
        (source: logits :181 and features :192; target: features :193 and logits :211) -- same
        values, 10 aggregations and two layer-0 projections fewer per step."""
        if (not self.adv and self.mode == 'node' and source_data.x.is_cuda
                and (getattr(source_data, "n_id", None) is not None or getattr(target_data, "n_id", None) is not None)):
            # sampled mini-batches (eager, host-bound): the MMD's draws + selection CSRs + staging copy are prepared on
            # a helper thread beside the forward passes (utils.mmd.prefetch_samples: same draws, same order)
            from ..utils.mmd import prefetch_samples
            prefetch_samples(source_data.x.size(0), target_data.x.size(0), source_data.x.device)
        from .. import ops as _ops
        # sampled steps of THIS trainer: every activation output feeds exactly one op (features -> the MMD, the stacked
        # second half -> the classifier's projection, the target's layer-0 draw -> the next projection), so the
        # activations may hand their backward to those producers' epilogues (ops.GradSink)
        sinks = (type(self) is A2GNN and not self.adv and self.mode == 'node' and source_data.x.is_cuda
                 and getattr(source_data, "n_id", None) is not None and getattr(target_data, "n_id", None) is not None)
        with (_ops.grad_sinks() if sinks else _null()):
            loss, source_logits, source_features, target_features, h0_t, pending, (sb, tb) = \
                self._branches(source_data, target_data)
        net = self.a2gnn
        loss = self._domain_loss(loss, source_features, target_features, alpha)
        if pending is not None:
            target_logits, side = pending
            torch.cuda.current_stream().wait_stream(side)                                # join
        elif self.compute_target_logits:                                                 # :211
            feats_t = net.feat_bottleneck_from(h0_t, target_data.edge_index, tb, self.t_pnums, draw=1)
            target_logits = net.feat_classifier(feats_t, target_data.edge_index, tb, 1)
        else:
            target_logits = None
        return loss, source_logits, target_logits

    def _dp_graph_parts(self):
        """The MMD step cut at its all-gather (pygda_amd/hipgraph.py::GraphedStepDP): everything up
        to the local row samples, and the global-batch MMD on the gathered rows.  None for the
        adversarial objective: it has no row exchange to cut at (its domain loss is a mean over local
        rows), so that data-parallel step runs eagerly."""
        if self.adv:
            return None
        from ..ops import mmd_loss_rows, sample_rows

        def part1(src, tgt, idx_s, idx_t, sel_s=None, sel_t=None):
            loss_ce, source_logits, sf, tf, _, pending, _ = self._branches(src, tgt)
            rows_s, rows_t = sample_rows(sf, idx_s, sel_s), sample_rows(tf, idx_t, sel_t)
            if pending is not None:
                torch.cuda.current_stream().wait_stream(pending[1])
            return loss_ce, source_logits, rows_s, rows_t

        def part2(rows_s_all, rows_t_all):
            return mmd_loss_rows(rows_s_all, rows_t_all) * self.weight

        return part1, part2

    def _prepare(self, source_data, target_data):
        """Everything fit() does before its epoch loop (a2gnn.py:254-296)."""
        # sampled mini-batches with the MMD loss: the step reads no per-epoch scalar and every batch of the device
        # sampler fits one capacity shape -- it may be captured once and replayed (pygda_amd/sampled_graph.py)
        self._sampled_graph_ok = type(self) is A2GNN and not self.adv and self.mode == 'node'
        self._loaders(source_data, target_data)                                          # :254-288
        self.a2gnn = self.init_model(**self.kwargs)
        # the MMD branch never reads alpha/epoch; the adversarial branch reads the GRL alpha, which the
        # captured step receives as a 0-dim device tensor refreshed per epoch: both replay as a hipGraph
        # (graph mode re-collates a shuffled batch every epoch: nothing static to capture)
        self._graph_safe_step, self._graph_uses_scalars = self.mode == 'node', bool(self.adv)
        # the MMD step reads no per-epoch scalar: consecutive steps may share one capture (hipgraph.GraphedStep.unroll)
        self._graph_unroll_ok = not self.adv and type(self) is A2GNN
        self._graph_forks = self.overlap_streams          # three parallel branches: the statistics ride on a fourth
        on_gpu = torch.device(self.device).type == "cuda"
        self._grad_aliases = None
        if on_gpu:           # torch.optim.Adam's update in one multi-tensor launch (pygda_amd/optim.py)
            from ..optim import Adam
            optimizer = Adam(self.a2gnn.parameters(), lr=self.lr, weight_decay=self.weight_decay)
            import os
            from .. import distributed
            # single process: the target branch's gradients of the shared layers go to second leaves and are summed inside
            # the update kernel (A2GNNBase.second_leaves); data-parallel runs average `.grad` and keep autograd's sums
            if (type(self) is A2GNN and os.environ.get("PYGDA_AMD_SECOND_LEAVES", "1") == "1"
                    and not distributed.active()):
                self._grad_aliases = optimizer.grad_aliases
        else:
            optimizer = torch.optim.Adam(self.a2gnn.parameters(), lr=self.lr, weight_decay=self.weight_decay)

        def step(src, tgt, alpha, epoch):
            loss, source_logits, _ = self.forward_model(src, tgt, alpha)
            return loss, source_logits

        def alpha(e):                                                                    # :305-306
            return 2. / (1. + np.exp(-10. * float(e) / self.epoch)) - 1

        return self.a2gnn, optimizer, step, alpha

    def fit(self, source_data, target_data):
        self._train_epochs(*self._prepare(source_data, target_data))

    def process_graph(self, data):
        pass

    def predict(self, data, source=False, reference_compat=None):
        """``data`` is ignored, as in the reference: the loaders stored by fit() are replayed.  ``reference_compat``
        (keyword, not in the reference): with several batches return what a2gnn.py:402-409 literally return instead of
        every node once -- see ``BaseGDA._predict_loader``; None = the trainer's ``reference_predict`` setting."""
        self.a2gnn.eval()
        loader, k = (self.source_loader, self.s_pnums) if source else (self.target_loader, self.t_pnums)
        return self._predict_loader(loader, lambda batch: self.a2gnn(batch, k), reference_compat)
