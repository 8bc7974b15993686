"""``BaseGDA`` (pygda/models/base.py:13-162): hyper-parameter container and the abstract
trainer API, plus the step loop every trainer of the reference re-implements by hand
(loaders -> zip -> forward_model -> Adam step -> per-epoch micro-F1 + log line)."""
import time
from abc import ABC, abstractmethod

import torch
import torch.nn.functional as F

from ..data import NeighborLoader
from ..metrics import eval_micro_f1
from ..utils import logger


class BaseGDA(ABC):
    def __init__(self, in_dim, hid_dim, num_classes, num_layers=2, dropout=0., weight_decay=0.,
                 act=F.relu, lr=4e-3, epoch=100, device='cuda:0', batch_size=0, num_neigh=-1,
                 verbose=2, **kwargs):
        self.in_dim, self.hid_dim, self.num_classes = in_dim, hid_dim, num_classes
        self.num_layers, self.dropout, self.weight_decay = num_layers, dropout, weight_decay
        self.act, self.verbose, self.kwargs = act, verbose, kwargs
        self.lr, self.epoch, self.device, self.batch_size = lr, epoch, device, batch_size
        if type(num_neigh) is int:                                   # base.py:86-95
            self.num_neigh = [num_neigh] * self.num_layers
        elif type(num_neigh) is list:
            if len(num_neigh) != self.num_layers:
                raise ValueError('Number of neighbors should have the '
                                 'same length as hidden layers dimension or'
                                 'the number of layers.')
            self.num_neigh = num_neigh
        else:
            raise ValueError('Number of neighbors must be int or list of int')
        self.model = None
        self.epoch_hook = None        # optional callable(epoch, loss, acc, seconds): bench / tests
        # capture the full-batch training step into a hipGraph (pygda_amd/hipgraph.py);
        # None = decide from the environment variable PYGDA_AMD_HIPGRAPH (default on: a step whose
        # capture fails falls back to eager launches with a warning)
        self.use_hip_graph = kwargs.pop("use_hip_graph", None)
        # tests: route even a whole-graph request (fan-out -1, batch_size >= N) through the sampler
        self.force_sampler = kwargs.pop("force_sampler", False)
        # predict() over SEVERAL batches: False (default) = every node once, the seeds' rows of every batch in loader order;
        # True = what the reference's loop literally returns (a2gnn.py:402-409, the same lines in every trainer): the
        # LAST batch's whole-batch logits twice beside the labels of all batches.  One batch: identical either way.
        # (``reference_predict=True`` here, ``predict(..., reference_compat=True)`` per call, or PYGDA_AMD_REFERENCE_PREDICT=1)
        import os
        self.reference_predict = bool(kwargs.pop("reference_predict", os.environ.get("PYGDA_AMD_REFERENCE_PREDICT", "0") == "1"))
        self.kwargs = kwargs

    # -- API of the reference -------------------------------------------------------
    def fit(self, data, **kwargs):
        """Subclasses override (base.py:99-111)."""

    def predict(self, data, **kwargs):
        """Subclasses override (base.py:113-125)."""

    @abstractmethod
    def init_model(self, **kwargs):
        ...

    @abstractmethod
    def process_graph(self, data, **kwargs):
        ...

    @abstractmethod
    def forward_model(self, data, **kwargs):
        ...

    # -- shared machinery -------------------------------------------------------------
    def _node_loaders(self, source_data, target_data):
        """a2gnn.py:254-277 (same block in every trainer): full batch when batch_size == 0."""
        self.num_source_nodes, self.num_target_nodes = source_data.x.shape[0], target_data.x.shape[0]
        dist = _dist_info()
        if self.batch_size == 0:
            self.source_batch_size, self.target_batch_size = self.num_source_nodes, self.num_target_nodes
            sb, tb = self.source_batch_size, self.target_batch_size
        else:
            sb = tb = self.batch_size
        full = self.batch_size == 0
        kw = (dict(auto_reorder=bool(getattr(self, "_auto_reorder_ok", False))) if full else
              dict(dist, device=self.device, full_batch=False if self.force_sampler else None, recycle=True))
        self.source_loader = NeighborLoader(source_data, self.num_neigh, batch_size=sb, **kw)
        self.target_loader = NeighborLoader(target_data, self.num_neigh, batch_size=tb, **kw)
        if not full:
            self._declare_static_shape(self.source_loader, self.target_loader)
        # MMD draws stay in the caller's numbering: the maps live on THIS trainer and are installed in utils.mmd only
        # while its own epoch loop runs (_train_epochs), never between fits or for another model
        maps = tuple(None if l.new_id is None else l.new_id.cpu() for l in (self.source_loader, self.target_loader))
        self._mmd_row_maps = maps if any(m is not None for m in maps) else None

    def _wants_sampled_graph(self):
        """A trainer whose sampled step may be captured at a static shape (pygda_amd/sampled_graph.py) says so with
        ``_sampled_graph_ok``; single process, on the GPU, not switched off."""
        import os
        from ..distributed import active
        return (getattr(self, "_sampled_graph_ok", False) and self.use_hip_graph is not False
                and os.environ.get("PYGDA_AMD_SAMPLED_GRAPH", "1") == "1" and os.environ.get("PYGDA_AMD_HIPGRAPH", "1") == "1"
                and torch.device(self.device).type == "cuda" and torch.cuda.is_available() and not active())

    def _declare_static_shape(self, *loaders):
        """Before the loaders' rings exist: every batch declares the interior capacity's worth of leading rows interior,
        so that all batches of a loader share ONE shape (sampled_graph.static_shape_ok)."""
        if not self._wants_sampled_graph():
            return
        from ..sampled_graph import static_shape_ok
        caps = [static_shape_ok(l) for l in loaders]
        if all(c is not None for c in caps):
            for l, c in zip(loaders, caps):
                l.static_interior = c

    def _sampled_stepper(self, net, optimizer, step_fn, capture=True):
        """The captured static-shape step for this trainer's sampled loaders, or None."""
        if not self._wants_sampled_graph():
            return None
        ls = (self.source_loader, self.target_loader)
        if any(getattr(l, "static_interior", 0) <= 0 or getattr(l, "full_batch", True) for l in ls):
            return None
        if not any(g.get("capturable", False) for g in optimizer.param_groups):
            return None
        hit = getattr(self, "_sampled_graphed", None)
        if hit is not None and hit[0] == (id(optimizer), id(ls[0]), id(ls[1]), capture):
            return hit[1]
        from ..sampled_graph import GraphedSampledStep
        stepper = GraphedSampledStep(self, net, step_fn, optimizer, ls[0], ls[1], capture=capture)
        self._sampled_graphed = ((id(optimizer), id(ls[0]), id(ls[1]), capture), stepper)
        return stepper

    def _graph_loaders(self, source_data, target_data):
        """``mode='graph'`` (a2gnn.py:278-286, the same block in grade.py:244-252, udagcn.py:248-256, adagcn.py:244-252,
        dane.py:219-229): ``DataLoader(dataset, batch_size, shuffle=True)`` over a list of graphs, every graph of the
        domain in one batch when ``batch_size == 0``."""
        from ..data import DataLoader
        bs_s = len(source_data) if self.batch_size == 0 else self.batch_size
        bs_t = len(target_data) if self.batch_size == 0 else self.batch_size
        self.source_loader = DataLoader(source_data, batch_size=bs_s, shuffle=True)
        self.target_loader = DataLoader(target_data, batch_size=bs_t, shuffle=True)
        self._mmd_row_maps = None

    def _loaders(self, source_data, target_data):
        mode = getattr(self, "mode", "node")
        if mode == 'node':
            self._node_loaders(source_data, target_data)
        elif mode == 'graph':
            self._graph_loaders(source_data, target_data)
        else:
            assert mode in ('graph', 'node'), 'Invalid train mode'

    def _train_epochs(self, net, optimizer, step_fn, alpha_fn, before_step=None, epochs=None):
        """The epoch loop of a2gnn.py:298-336: ``step_fn(src, tgt, alpha, epoch)`` returns
        ``(loss, source_logits)``; the optimiser step happens here.  ``epochs`` (default
        ``range(self.epoch)``) lets a harness run the same loop in slices."""
        from ..utils.mmd import scoped_row_maps
        with scoped_row_maps(getattr(self, "_mmd_row_maps", None)):
            return self._train_epochs_scoped(net, optimizer, step_fn, alpha_fn, before_step, epochs)

    def _train_epochs_scoped(self, net, optimizer, step_fn, alpha_fn, before_step=None, epochs=None):
        start = time.time()
        _freeze_gc()
        if not getattr(self, "_dp_synced", None) is net:      # data-parallel: one set of initial weights
            from ..distributed import broadcast_parameters, direct_agreed
            direct_agreed()           # nccl groups: all ranks on the library-owned RCCL communicator, or all on the ProcessGroup
            broadcast_parameters(net)
            for aux in getattr(self, "_dp_aux_modules", ()):      # critics / discriminators with their own optimiser
                broadcast_parameters(aux)
            self._dp_synced = net
        graphed = self._maybe_graphed_step(optimizer, step_fn, before_step, net)
        if graphed is not None and hasattr(graphed, "launch"):
            return self._graphed_epochs(graphed, range(self.epoch) if epochs is None else epochs, start, alpha_fn)
        import os
        # (PYGDA_AMD_SAMPLED_GRAPH_CAPTURE=0: the same static-shape step issued eagerly every time -- tests)
        stepper = self._sampled_stepper(net, optimizer, step_fn,
                                        capture=os.environ.get("PYGDA_AMD_SAMPLED_GRAPH_CAPTURE", "1") == "1") \
            if (graphed is None and before_step is None) else None
        for epoch in (range(self.epoch) if epochs is None else epochs):
            epoch_loss, logits, labels, dev_loss = 0.0, [], [], None
            alpha = alpha_fn(epoch)
            raw = None
            if stepper is not None:
                raw = (self.source_loader.iter_raw(), self.target_loader.iter_raw())
                if raw[0] is None or raw[1] is None:
                    raw = stepper = None
            if raw is not None:
                # sampled mini-batches, the step replayed at its static shape (a2gnn.py:308-319 per pair; same numbers):
                # per step the captured graph leaves {loss, #correct source rows}; they are read a few steps later, so the
                # host never waits for the step it has just launched
                tickets, rows, correct, dev_correct = [], 0, 0.0, None
                S_s, S_t = self.source_loader._sampler, self.target_loader._sampler

                def settle(keep):
                    nonlocal epoch_loss, rows, correct
                    while len(tickets) > keep:
                        loss_v, corr, n_live = stepper.result(tickets.pop(0))
                        epoch_loss += loss_v
                        correct += corr
                        rows += n_live

                for (ps, zs), (pt, zt) in zip(*raw):
                    ticket = stepper.step(ps, zs, pt, zt)
                    if ticket is not None:
                        tickets.append(ticket)
                        settle(2)
                        continue
                    # a pair the static shape cannot take: the ordinary eager step on its real shape
                    net.train()
                    src, tgt = S_s.assemble(self.source_loader.data, ps, zs), S_t.assemble(self.target_loader.data, pt, zt)
                    from ..ops import dropout_state
                    dropout_state.next_step(src.x.device)
                    loss, source_logits = step_fn(src, tgt, alpha, epoch)
                    optimizer.zero_grad()
                    loss.backward()
                    optimizer.step()
                    dev_loss = loss.detach().double() if dev_loss is None else dev_loss + loss.detach().double()
                    hit = (source_logits.detach().argmax(dim=1) == src.y).sum()
                    dev_correct = hit if dev_correct is None else dev_correct + hit
                    rows += int(src.y.numel())
                    S_s.release(ps)              # everything that reads the two ring blocks is enqueued: hand them back
                    S_t.release(pt)
                settle(0)
                for gen in raw:              # (a zip that stopped at the shorter loader leaves the other generator open)
                    gen.close()
                if dev_loss is not None:
                    epoch_loss += dev_loss.item()
                    correct += float(dev_correct.item())
                acc = correct / rows if rows else 0.0
                secs = time.time() - start
                logger(epoch=epoch, loss=epoch_loss, source_train_acc=acc, time=secs, verbose=self.verbose, train=True)
                if self.epoch_hook is not None:
                    self.epoch_hook(epoch, epoch_loss, acc, secs)
                continue
            if graphed is not None:
                loss, source_logits = graphed()
                epoch_loss += loss.item()
                logits.append(source_logits)
                labels.append(graphed.src.y)
            for src, tgt in (() if graphed is not None else zip(self.source_loader, self.target_loader)):
                if before_step is not None:
                    before_step()
                else:
                    net.train()
                src, tgt = src.to(self.device), tgt.to(self.device)
                if src.x.is_cuda:
                    from ..ops import dropout_state
                    dropout_state.next_step(src.x.device)      # fresh masks for the fused activations
                loss, source_logits = step_fn(src, tgt, alpha, epoch)
                optimizer.zero_grad()
                loss.backward()
                _allreduce_grads(optimizer)
                optimizer.step()
                # the reference adds loss.item() per batch (a2gnn.py:327): a host sync per step, with which the host can
                # never run ahead of the device.  Same doubles, summed on the device in step order, read once per epoch
                dev_loss = loss.detach().double() if dev_loss is None else dev_loss + loss.detach().double()
                logits.append(source_logits.detach())
                labels.append(src.y)
            if dev_loss is not None:
                epoch_loss += dev_loss.item()
            preds = torch.cat(logits).argmax(dim=1)
            acc = eval_micro_f1(torch.cat(labels), preds)
            secs = time.time() - start
            logger(epoch=epoch, loss=epoch_loss, source_train_acc=acc, time=secs,
                   verbose=self.verbose, train=True)
            if self.epoch_hook is not None:
                self.epoch_hook(epoch, epoch_loss, acc, secs)

    def _stacked_pair(self, source_data, target_data):
        """Two full-batch graphs as ONE block-diagonal batch -- rows ``[source nodes ; target nodes]``, the target's edges
        shifted by the source's node count -- for trainers that run the SAME network over both domains
        (pygda/models/grade.py:162-163 and the like): one pass over ``ns + nt`` rows instead of two passes, half the
        launches each way on a step that is launch-latency bound, one weight-gradient product per layer instead of two
        and an accumulation.  Per-row results are the two passes' (same neighbours in the same order, the same
        normalisation, row-wise layers).  Built once per pair of static device graphs (bag-of-words features registered
        for the sparse projection like any feature matrix entering the device); None when the batches are not such a pair."""
        import os
        from ..data import Data
        s, t = source_data, target_data
        if (os.environ.get("PYGDA_AMD_STACKED_DOMAINS", "1") != "1" or self.mode != 'node'
                or getattr(s, "x", None) is None or not s.x.is_cuda or s.x.dim() != 2 or t.x.dim() != 2
                or s.x.size(1) != t.x.size(1) or s.x.dtype != t.x.dtype
                or not getattr(s.edge_index, "_gda_static", False) or not getattr(t.edge_index, "_gda_static", False)
                or getattr(s, "edge_weight", None) is not None or getattr(t, "edge_weight", None) is not None):
            return None
        key = (s.x.data_ptr(), t.x.data_ptr(), s.edge_index.data_ptr(), t.edge_index.data_ptr(), s.x._version, t.x._version)
        hit = self.__dict__.get("_stacked_pair_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        ns = s.x.size(0)
        with torch.no_grad():
            x = torch.cat([s.x, t.x])
            ei = torch.cat([s.edge_index, t.edge_index + ns], dim=1)
        ei._gda_static = True
        from .. import sparse_features
        sparse_features.maybe_register(x)
        both = Data(x=x, edge_index=ei, y=None)
        both._static_graph = True
        both.ns = ns
        self._stacked_pair_cache = (key, both, s, t)          # (keeps the two batches alive: the key holds their addresses)
        return both

    def _graphed_epochs(self, graphed, epochs, start, alpha_fn=None):
        """Full-batch epochs as hipGraph replays, software-pipelined by one step: the host draws the
        MMD samples of epoch e+1 and launches it while epoch e's two numbers (loss, source accuracy --
        micro-F1 of single-label predictions -- computed inside the graph) travel back, so the GPU
        never waits for the log line.  Same values, same order, reported one launch later."""
        def report(epoch, ticket):
            loss, acc = graphed.result(ticket)
            secs = time.time() - start
            logger(epoch=epoch, loss=loss, source_train_acc=acc, time=secs, verbose=self.verbose, train=True)
            if self.epoch_hook is not None:
                self.epoch_hook(epoch, loss, acc, secs)

        def report_group(group, ticket):
            if isinstance(ticket, tuple):                  # several steps behind one replay (GraphedStep.unroll)
                for epoch, (loss, acc) in zip(group, graphed.result_multi(ticket)):
                    secs = time.time() - start
                    logger(epoch=epoch, loss=loss, source_train_acc=acc, time=secs, verbose=self.verbose, train=True)
                    if self.epoch_hook is not None:
                        self.epoch_hook(epoch, loss, acc, secs)
            else:
                report(group[0], ticket)

        scalars = alpha_fn is not None and getattr(self, "_graph_uses_scalars", False)
        unroll = 1 if scalars or getattr(graphed, "graph_multi", None) is None else graphed.unroll
        epochs = list(epochs)
        pending, i = None, 0
        while i < len(epochs):
            if unroll > 1 and len(epochs) - i >= unroll:
                group, ticket = epochs[i:i + unroll], graphed.launch_multi()
            else:
                if scalars:
                    self._g_alpha.fill_(float(alpha_fn(epochs[i])))
                    self._g_epoch.fill_(float(epochs[i]))
                group, ticket = epochs[i:i + 1], graphed.launch()
            i += len(group)
            if pending is not None:
                report_group(*pending)
            pending = (group, ticket)
        if pending is not None:
            report_group(*pending)

    def _maybe_graphed_step(self, optimizer, step_fn, before_step, net):
        """A captured step when asked for and legal: one full-batch pair, single process, and a
        step whose arithmetic does not depend on per-epoch Python scalars."""
        import os
        want = self.use_hip_graph
        if want is None:
            want = os.environ.get("PYGDA_AMD_HIPGRAPH", "1") == "1"
        if not want or not getattr(self, "_graph_safe_step", False):
            return None
        if getattr(self, "_graphed", None) is not None and self._graphed_key == id(optimizer):
            return self._graphed
        from ..distributed import active
        if not torch.cuda.is_available():
            return None
        from ..distributed import capture_collectives
        dp = active()
        whole = dp and capture_collectives()          # library-owned RCCL communicator: collectives capture (opt-in)
        parts = self._dp_graph_parts() if (dp and not whole and hasattr(self, "_dp_graph_parts")) else None
        if dp and not whole and parts is None:
            return None       # RCCL collectives abort under stream capture on this stack (ROCm 7.0 /
                              # torch 2.10): without a segmented step, data-parallel training stays eager
        if not (getattr(self.source_loader, "full_batch", False) and getattr(self.target_loader, "full_batch", False)):
            return None
        extra = getattr(self, "_graph_extra_optimizers", ())
        if not all(any(g.get("capturable", False) for g in o.param_groups) for o in (optimizer, *extra)):
            return None
        from ..hipgraph import GraphedStep, GraphedStepDP
        # A trainer and its captured graphs form a reference cycle, so an earlier, dropped trainer's
        # hipGraphs die whenever the cyclic collector runs -- destroying a graph (and its memory pool) in
        # the middle of this trainer's replays has crashed the runtime.  Collect them now, device idle.
        import gc
        torch.cuda.synchronize()
        gc.collect()
        src = next(iter(self.source_loader)).to(self.device)
        tgt = next(iter(self.target_loader)).to(self.device)
        (before_step or net.train)()
        params = [p for o in (optimizer, *extra) for g in o.param_groups for p in g["params"]]
        saved = [p.detach().clone() for p in params]
        cpu_rng = torch.get_rng_state()
        # per-epoch scalars of the reference's loops (GRL alpha, epoch) as 0-dim device tensors: refreshed
        # before every replay, so the captured arithmetic sees the current value
        self._g_alpha = torch.zeros((), dtype=torch.float32, device=src.x.device)
        self._g_epoch = torch.zeros((), dtype=torch.float32, device=src.x.device)
        scalar_step = lambda s, t: step_fn(s, t, self._g_alpha, self._g_epoch)      # noqa: E731
        try:
            if whole:
                graphed = GraphedStep(scalar_step, optimizer, src, tgt, dp=True).capture()
            elif dp:          # collectives stay eager between four captured segments
                def eager_step():
                    from ..ops import dropout_state
                    dropout_state.next_step(src.x.device)
                    loss, _ = step_fn(src, tgt, 0.0, 0)
                    optimizer.zero_grad()
                    loss.backward()
                    _allreduce_grads(optimizer)
                    optimizer.step()
                part1, part2 = parts
                graphed = GraphedStepDP(part1, part2, optimizer, src, tgt).capture(eager_step)
            else:
                split = self._split_graph_parts() if hasattr(self, "_split_graph_parts") else None
                if split is not None and not getattr(self, "_graph_extra_optimizers", ()):
                    from ..hipgraph import GraphedStepSplit
                    graphed = GraphedStepSplit(split, self._g_alpha, scalar_step, optimizer, src, tgt).capture()
                else:
                    # several steps per replay: the weights are already U-1 steps (plus the pipelining lag) past
                    # the epoch a log line reports.  A hook may want to look at the MODEL of that epoch (evaluate,
                    # checkpoint), so a trainer with an epoch_hook replays one step at a time unless the
                    # environment variable asks for more explicitly; hooks should consume the numbers they are
                    # handed -- with pipelined epochs the model is always at least one launch ahead of them
                    env_unroll = os.environ.get("PYGDA_AMD_GRAPH_UNROLL")
                    # (four steps per replay since the refill of a multi-step replay is ONE copy: hipgraph.GraphedStep._provider)
                    unroll = int(env_unroll or ("1" if self.epoch_hook is not None else "4")) \
                        if getattr(self, "_graph_unroll_ok", False) else 1
                    graphed = GraphedStep(scalar_step, optimizer, src, tgt,
                                          extra_optimizers=getattr(self, "_graph_extra_optimizers", ()),
                                          unroll=unroll,
                                          inline_stats=not getattr(self, "_graph_forks", False)
                                          and os.environ.get("PYGDA_AMD_INLINE_STATS", "1") == "1").capture()
        except Exception as exc:       # anything a custom activation / exotic configuration may do under capture
            if self.use_hip_graph:     # explicitly requested: do not hide the failure
                raise
            import warnings
            warnings.warn(f"hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); "
                          "falling back to eager launches")
            torch.cuda.synchronize()
            with torch.no_grad():      # undo whatever the warm-up steps did
                for p, v in zip(params, saved):
                    p.copy_(v)
                    p.grad = None
                for o in (optimizer, *extra):
                    if hasattr(o, "_bumped"):
                        o._bumped = None      # a bump whose step() never came (optim.Adam.bump_steps) must not leak into eager steps
                    for st in o.state.values():
                        for v in st.values():
                            if torch.is_tensor(v):
                                v.zero_()
            torch.set_rng_state(cpu_rng)
            self._graph_safe_step = False
            return None
        self._graphed = graphed
        self._graphed_key = id(optimizer)
        return self._graphed

    def _gmean(self, mean_local, n_local):
        """A loss that is a mean over ``n_local`` rows of this rank's batch -> the mean over every rank's
        rows (sub-graphs differ in size, so CE means are weighted by node counts).  Identity in
        single-process runs and for full-batch replicas, whose counts are equal by construction."""
        from ..distributed import active, global_mean
        if not active() or getattr(getattr(self, "source_loader", None), "full_batch", False):
            return mean_local
        return global_mean(mean_local, n_local)

    def _predict_loader(self, loader, forward, reference_compat=None):
        """predict() of the reference (a2gnn.py:384-411) over the loader fit() stored.

        One batch (every benchmark setting of the reference): that batch's logits and labels, as in the reference.
        Several batches, default: every node exactly once -- the seeds' rows of every batch, in loader order (what the
        docstring of the reference's predict() describes: "concatenates results for full predictions").
        Several batches, ``reference_compat=True``: what the reference's loop RETURNS -- its ``idx > 0`` branch assigns
        the fresh ``logits`` before concatenating "the accumulated logits" with it (:402-405, :413-416), so the result is
        the LAST batch's whole-batch logits (seeds and sampled neighbours) twice, beside the labels of ALL batches
        (whole batches): ``logits.shape[0] != labels.shape[0]`` in general.  Tested against a reference-run golden
        (``a2gnn_fit3_mb_*.npz``); the default's deviation is deliberate and is exactly this."""
        compat = self.reference_predict if reference_compat is None else bool(reference_compat)
        outs, labs = [], []
        with torch.no_grad():
            for batch in loader:
                batch = batch.to(self.device)
                out = forward(batch)
                k = None if compat else getattr(batch, "batch_size", None)
                if compat:
                    outs = [out]                  # :402 / :413 -- `logits = self.a2gnn(...)` replaces what was accumulated
                    labs.append(batch.y)
                else:
                    outs.append(out if k is None else out[:k])
                    labs.append(batch.y if k is None else batch.y[:k])
        if compat and len(labs) > 1:
            out, lab = torch.cat((outs[0], outs[0])), torch.cat(labs)        # :405 / :416
        else:
            out, lab = torch.cat(outs), torch.cat(labs)
        new_id = getattr(loader, "new_id", None)
        if new_id is not None:                   # the loader trains on a degree-ordered relabelling (data.auto_reorder):
            new_id = new_id.to(out.device)       # row new_id[i] of its batch is node i of the caller's numbering
            out, lab = out[new_id], lab[new_id]
        return out, lab


_gc_frozen = False


def _freeze_gc():
    """Once per process, before the first training loop: collect, then move everything alive -- the interpreter, torch
    and its ~10^6 functions, types and modules -- into the cyclic collector's permanent generation.  A full collection
    otherwise walks all of it: 70 ms on the GPU box = twenty sampled-training steps lost whenever generation 2 comes
    up (measured: steps of 3.1 ms with one of 71 ms and one of 9.5 ms per ~100; none with the collector off).  Objects
    created afterwards are collected as usual.  ``PYGDA_AMD_GC_FREEZE=0`` leaves the collector alone."""
    global _gc_frozen
    import gc
    import os
    if _gc_frozen or os.environ.get("PYGDA_AMD_GC_FREEZE", "1") != "1":
        return
    gc.collect()
    gc.freeze()
    _gc_frozen = True


def _dist_info():
    from ..distributed import info
    return info()


def _allreduce_grads(optimizer):
    """Data-parallel step: ONE flat all-reduce (RCCL over xGMI when the backend is nccl) of
    all gradients, averaged over ranks.  No-op in single-process runs."""
    from ..distributed import allreduce_grads
    allreduce_grads(p for g in optimizer.param_groups for p in g["params"])
