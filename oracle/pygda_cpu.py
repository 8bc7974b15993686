"""CPU oracle: a plain-torch fp32 restatement of pygda's A2GNN-style GDA training path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``pygda_amd/`` imports it, and the product path raises if the HIP
library is missing instead of falling back to this file.

Every function cites the reference ``file:line`` (relative to the reference
checkout of pygda-team/pygda v1.2.1) whose arithmetic it restates, op for op and
in the same evaluation order, so that on CPU the results are bit-identical to the
reference running on PyG's non-fused path (``index_select`` -> multiply ->
scatter-add in edge order).

Pinning status (see ``tests/golden/make_golden.py`` and profiles/HISTORY.md §3):
  * ``guassian_kernel`` / ``get_MMD`` / ``MMD`` / ``GradReverse`` / ``Attention``:
    PINNED -- checked against the reference's own files imported by path.
  * ``gcn_norm`` / ``PropGCNConv`` / ``CachedGCNConv`` / ``A2GNNBase`` /
    ``A2GNN.forward_model`` / GRADE / UDAGCN / AdaGCN restatements: checked against the
    reference's own files executed on a build-authored stand-in for the PyG calls they
    make (PyG, torch_scatter, torch_sparse are absent from the reference checkout and
    cannot be installed here; pinned versions: torch_geometric>=2.4.0,
    torch_scatter>=2.1.0, torch_sparse>=0.6.15, README.md:56-59).  The reference
    holds no tests or golden vectors at that boundary, so for these symbols parity is
    **UNPINNED at the PyG boundary**: the PyG semantics are restated from its
    published algorithm (listed in ``tests/golden/_pyg_stub.py``).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


# ----------------------------------------------------------------------------
# graph normalisation  (a1)
# ----------------------------------------------------------------------------
def add_remaining_self_loops(edge_index: Tensor, edge_weight: Tensor,
                             fill_value: float, num_nodes: int
                             ) -> Tuple[Tensor, Tensor]:
    """PyG ``add_remaining_self_loops`` as called from prop_gcn_conv.py:72 and
    cached_gcn_conv.py:95: existing loops are dropped from the edge list, one
    loop per node is appended LAST in node order; a node that already had a loop
    keeps that loop's weight (last writer in edge order wins)."""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop_w = torch.full((num_nodes,), float(fill_value), dtype=edge_weight.dtype)
    inv = ~mask
    # sequential last-writer-wins, as CPU index_put does
    loop_w[row[inv]] = edge_weight[inv]
    loop_idx = torch.arange(num_nodes, dtype=torch.long)
    ei = torch.cat([edge_index[:, mask], torch.stack([loop_idx, loop_idx])], dim=1)
    ew = torch.cat([edge_weight[mask], loop_w])
    return ei, ew


def gcn_norm(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
             improved: bool = False, add_self_loops: bool = True,
             degree_side: str = "col") -> Tuple[Tensor, Tensor]:
    """prop_gcn_conv.py:64-81 (degree over ``col``, the destination) and
    cached_gcn_conv.py:88-103 (``degree_side='row'``, the source)."""
    fill = 2.0 if improved else 1.0
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=torch.float32)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(
            edge_index, edge_weight, fill, num_nodes)
    row, col = edge_index[0], edge_index[1]
    idx = col if degree_side == "col" else row
    deg = torch.zeros(num_nodes, dtype=edge_weight.dtype).index_add_(0, idx, edge_weight)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return edge_index, dis[row] * edge_weight * dis[col]


def propagate(edge_index: Tensor, edge_weight: Tensor, x: Tensor) -> Tensor:
    """PyG ``MessagePassing.propagate`` with aggr='add', flow source->target, as
    used at prop_gcn_conv.py:209 (+ message :238) and cached_gcn_conv.py:138,156:
    ``out[i] = sum_{e: col[e]==i} w[e] * x[row[e]]`` accumulated in edge order."""
    msg = edge_weight.view(-1, 1) * x.index_select(0, edge_index[0])
    return torch.zeros(x.size(0), x.size(1), dtype=x.dtype).index_add_(0, edge_index[1], msg)


def glorot_(t: Tensor) -> Tensor:
    """PyG ``inits.glorot``: U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))."""
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)
    return t


# ----------------------------------------------------------------------------
# operators  (a2, a8, a10, a12)
# ----------------------------------------------------------------------------
class _Lin(nn.Module):
    """PyG ``Linear(in, out, bias=False, weight_initializer='glorot')``: weight
    ``[out, in]``, one glorot draw at construction (no torch kaiming draw)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        glorot_(self.weight)


class PropGCNConv(nn.Module):
    """prop_gcn_conv.py:84-215.  ``lin`` has no bias and weight ``[out, in]``; the
    glorot draw happens twice (PyG ``Linear.__init__`` and then
    ``PropGCNConv.reset_parameters`` :144-147), which matters only for RNG-stream
    parity of freshly initialised models."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False,
                 add_self_loops: bool = True, normalize: bool = True, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.add_self_loops, self.normalize = improved, add_self_loops, normalize
        self.lin = _Lin(in_channels, out_channels)   # PyG Linear.__init__ draws glorot once
        glorot_(self.lin.weight)          # PropGCNConv.reset_parameters
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, edge_index, prop_nums=1, edge_weight=None):
        if self.normalize:                                   # :182-192 (cached=False)
            edge_index, edge_weight = gcn_norm(edge_index, edge_weight, x.size(0),
                                               self.improved, self.add_self_loops, "col")
        elif edge_weight is None:
            edge_weight = torch.ones(edge_index.size(1), dtype=x.dtype)
        out = F.linear(x, self.lin.weight)                   # :205
        for _ in range(prop_nums):                           # :208-210
            out = propagate(edge_index, edge_weight, out)
        if self.bias is not None:                            # :212-213
            out = out + self.bias
        return out


class GCNConv(PropGCNConv):
    """PyG ``GCNConv`` as used by grade_base.py:58-61, adagcn_base.py:49-52,
    gnn_base.py:65-71: lin (glorot, no bias) -> gcn_norm(col) -> 1 propagate -> + bias."""

    def forward(self, x, edge_index, edge_weight=None):       # noqa: D102
        return super().forward(x, edge_index, 1, edge_weight)


class CachedGCNConv(nn.Module):
    """cached_gcn_conv.py:35-174: ``x @ W`` (W ``[in, out]``), source-side degree
    norm cached per ``cache_name``, one propagate, bias added in ``update``."""

    def __init__(self, in_channels, out_channels, weight=None, bias=None,
                 improved=False, use_bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.improved = in_channels, out_channels, improved
        self.cache_dict = {}
        if weight is None:
            self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
            glorot_(self.weight)
        else:
            self.weight = weight
        if bias is None:
            self.bias = nn.Parameter(torch.zeros(out_channels)) if use_bias else None
        else:
            self.bias = bias

    def forward(self, x, edge_index, cache_name="default_cache", edge_weight=None):
        x = torch.matmul(x, self.weight)                                    # :130
        if cache_name not in self.cache_dict:                               # :132-136
            self.cache_dict[cache_name] = gcn_norm(edge_index, edge_weight, x.size(0),
                                                   self.improved, True, "row")
        ei, norm = self.cache_dict[cache_name]
        out = propagate(ei, norm, x)                                        # :138,156
        if self.bias is not None:                                           # :172-174
            out = out + self.bias
        return out


class _GradReverse(torch.autograd.Function):
    """reverse_layer.py:4-66: identity forward, ``-alpha * g`` backward."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.neg() * ctx.alpha, None


def grad_reverse(x: Tensor, alpha: float) -> Tensor:
    return _GradReverse.apply(x, alpha)


class Attention(nn.Module):
    """attention.py:6-55: softmax(Linear(h,1)) over K stacked views, weighted sum.
    (The Dropout(0.1) the reference constructs at :26 is never applied.)"""

    def __init__(self, in_channels):
        super().__init__()
        self.dense_weight = nn.Linear(in_channels, 1)

    def forward(self, inputs: List[Tensor]) -> Tensor:
        stacked = torch.stack(inputs, dim=1)
        weights = F.softmax(self.dense_weight(stacked), dim=1)
        return torch.sum(stacked * weights, dim=1)


# ----------------------------------------------------------------------------
# MMD  (a7)
# ----------------------------------------------------------------------------
def guassian_kernel(source, target, kernel_mul=2.0, kernel_num=5, fix_sigma=None,
                    chunk_rows: Optional[int] = None):
    """mmd.py:4-55.  ``chunk_rows`` only bounds the size of the ``[n,n,d]``
    temporary (the per-element arithmetic and its order are unchanged); ``None``
    materialises it whole exactly as the reference does."""
    n = int(source.size(0)) + int(target.size(0))
    total = torch.cat([source, target], dim=0)
    if chunk_rows is None:
        t0 = total.unsqueeze(0).expand(n, n, total.size(1))
        t1 = total.unsqueeze(1).expand(n, n, total.size(1))
        L2 = ((t0 - t1) ** 2).sum(2)                                        # :43-46
    else:
        rows = []
        for s in range(0, n, chunk_rows):
            blk = total[s:s + chunk_rows]
            rows.append(((total.unsqueeze(0) - blk.unsqueeze(1)) ** 2).sum(2))
        L2 = torch.cat(rows, dim=0)
    if fix_sigma:
        bandwidth = fix_sigma
    else:
        bandwidth = (torch.sum(L2.data) + 1e-6) / (n ** 2 - n)              # :50
    bandwidth = bandwidth / (kernel_mul ** (kernel_num // 2))               # :51
    bw_list = [bandwidth * (kernel_mul ** i) for i in range(kernel_num)]    # :52
    return sum(torch.exp(-L2 / bw) for bw in bw_list)                       # :53-55


def get_MMD(source_feat, target_feat, kernel_mul=2.0, kernel_num=5, fix_sigma=None,
            chunk_rows: Optional[int] = None):
    """mmd.py:57-107."""
    k = guassian_kernel(source_feat, target_feat, kernel_mul, kernel_num, fix_sigma, chunk_rows)
    b = min(int(source_feat.size(0)), int(target_feat.size(0)))
    return torch.mean(k[:b, :b] + k[b:, b:] - k[:b, b:] - k[b:, :b])        # :100-106


def MMD(source_feat, target_feat, sampling_num=1000, times=5,
        chunk_rows: Optional[int] = None, samples=None):
    """mmd.py:109-159.  Row indices come from the CPU default generator
    (``torch.randint`` without a device, :148-149) unless ``samples`` (a pair of
    ``[times, sampling_num]`` int64 tensors) is supplied."""
    if samples is None:
        s_idx = torch.randint(source_feat.size(0), (times, sampling_num))
        t_idx = torch.randint(target_feat.size(0), (times, sampling_num))
    else:
        s_idx, t_idx = samples
    mmd = 0
    for i in range(times):
        mmd = mmd + get_MMD(source_feat[s_idx[i]], target_feat[t_idx[i]], chunk_rows=chunk_rows)
    return mmd / times


# ----------------------------------------------------------------------------
# backbones and forward_model restatements  (a3, a4, a9, a13, a14)
# ----------------------------------------------------------------------------
class Graph:
    """The reference's data contract: ``.x``, ``.edge_index``, ``.y``."""

    def __init__(self, x, edge_index, y=None):
        self.x, self.edge_index, self.y = x, edge_index, y


def global_mean_pool(x: Tensor, batch: Tensor, size: Optional[int] = None) -> Tensor:
    """PyG ``global_mean_pool`` (a2gnn_base.py:141 on a DataLoader's collated batch): per-graph sum of the node rows
    (scatter in node order) divided by the node count, clamped at 1."""
    n = int(batch.max()) + 1 if size is None else size
    total = torch.zeros(n, x.size(1), dtype=x.dtype).index_add_(0, batch, x)
    count = torch.zeros(n, dtype=x.dtype).index_add_(0, batch, torch.ones(batch.numel(), dtype=x.dtype))
    return total / count.clamp(min=1).view(-1, 1)


def collate_graphs(graphs) -> "Graph":
    """PyG ``Batch.from_data_list`` for the attributes graph mode reads (a2gnn.py:278-286 DataLoader batches):
    ``x`` / ``y`` concatenated in list order, ``edge_index`` shifted by the nodes before each graph, ``batch``."""
    counts = torch.tensor([g.x.size(0) for g in graphs], dtype=torch.long)
    offs = (torch.cumsum(counts, 0) - counts).tolist()
    out = Graph(torch.cat([g.x for g in graphs], 0),
                torch.cat([g.edge_index + o for g, o in zip(graphs, offs)], 1),
                torch.cat([g.y.reshape(-1) for g in graphs], 0))
    out.batch = torch.repeat_interleave(torch.arange(len(graphs)), counts)
    out.num_graphs = len(graphs)
    return out


class A2GNNBase(nn.Module):
    """a2gnn_base.py:11-203 (``mode='node'``; ``mode='graph'``: mean readout per graph :140-141, linear classifier)."""

    def __init__(self, in_dim, hid_dim, num_classes, num_layers=1, adv=False,
                 dropout=0.1, act=F.relu, mode="node"):
        super().__init__()
        self.dropout, self.act, self.adv, self.mode = dropout, act, adv, mode
        self.convs = nn.ModuleList([PropGCNConv(in_dim, hid_dim)])
        for _ in range(num_layers - 1):
            self.convs.append(PropGCNConv(hid_dim, hid_dim))
        self.cls = PropGCNConv(hid_dim, num_classes) if mode == "node" else nn.Linear(hid_dim, num_classes)
        if adv:
            self.domain_discriminator = nn.Linear(hid_dim, 2)

    def feat_bottleneck(self, x, edge_index, batch=None, prop_nums=30):      # :106-143
        for conv in self.convs:
            x = conv(x, edge_index, prop_nums)
            x = self.act(x)
            x = F.dropout(x, p=self.dropout, training=self.training)
        if self.mode == "graph":                                             # :140-141
            x = global_mean_pool(x, batch)
        return x

    def feat_classifier(self, x, edge_index, batch=None, prop_nums=1):       # :145-176
        return self.cls(x, edge_index, prop_nums) if self.mode == "node" else self.cls(x)

    def domain_classifier(self, x, alpha):                                   # :178-203
        return self.domain_discriminator(grad_reverse(x, alpha))

    def forward(self, data, prop_nums):                                      # :72-104
        batch = None if self.mode == "node" else data.batch
        x = self.feat_bottleneck(data.x, data.edge_index, batch, prop_nums)
        return self.feat_classifier(x, data.edge_index, batch, 1)


def a2gnn_forward_model(net: A2GNNBase, src: Graph, tgt: Graph, alpha: float,
                        s_pnums: int, t_pnums: int, adv: bool, weight: float,
                        mmd_chunk_rows: Optional[int] = None, mmd_samples=None):
    """a2gnn.py:146-213: CE(source) + weight * (MMD | GRL domain CE); the second
    target forward (:211) is returned but not part of the loss."""
    source_logits = net(src, s_pnums)                                        # :181
    loss = F.nll_loss(F.log_softmax(source_logits, dim=1), src.y)            # :182
    sb, tb = getattr(src, "batch", None), getattr(tgt, "batch", None)        # :186-190 (None in node mode)
    sf = net.feat_bottleneck(src.x, src.edge_index, sb, s_pnums)             # :192
    tf = net.feat_bottleneck(tgt.x, tgt.edge_index, tb, t_pnums)             # :193
    if adv:                                                                  # :196-205
        sd = net.domain_classifier(sf, alpha)
        td = net.domain_classifier(tf, alpha)
        lab = torch.tensor([0] * src.x.shape[0] + [1] * tgt.x.shape[0])
        loss = loss + weight * F.cross_entropy(torch.cat([sd, td], 0), lab)
    else:                                                                    # :206-209
        loss = loss + MMD(sf, tf, chunk_rows=mmd_chunk_rows, samples=mmd_samples) * weight
    target_logits = net(tgt, t_pnums)                                        # :211
    return loss, source_logits, target_logits


class GRADEBase(nn.Module):
    """grade_base.py:9-202 (``mode='graph'``: every layer's output and the last one mean-pooled per graph, :150-157)."""

    def __init__(self, in_dim, hid_dim, num_classes, num_layers=1, dropout=0.1,
                 act=F.relu, disc="JS", mode="node"):
        super().__init__()
        self.dropout, self.act, self.num_classes, self.mode = dropout, act, num_classes, mode
        self.convs = nn.ModuleList([GCNConv(in_dim, hid_dim)])
        for _ in range(num_layers - 1):
            self.convs.append(GCNConv(hid_dim, hid_dim))
        self.cls = nn.Linear(hid_dim, num_classes)
        width = hid_dim * num_layers + num_classes * (1 if disc == "JS" else 2)
        self.discriminator = nn.Sequential(nn.Linear(width, 2))

    def forward(self, data):                                                 # :78-115
        x, feats = data.x, []
        for conv in self.convs:
            x = conv(x, data.edge_index)
            x = self.act(x)
            x = F.dropout(x, p=self.dropout, training=self.training)
            feats.append(x if self.mode == "node" else global_mean_pool(x, data.batch))       # :150-153
        if self.mode == "graph":                                                               # :155-156
            x = global_mean_pool(x, data.batch)
        x = self.cls(x)
        feats.append(x)
        return x, torch.cat(feats, dim=1)


def grade_forward_model(net: GRADEBase, src: Graph, tgt: Graph, alpha: float,
                        disc: str, weight: float, mmd_chunk_rows=None, mmd_samples=None):
    """grade.py:129-197 for disc in {'JS', 'MMD', 'C'}."""
    s_logits, s_feats = net(src)
    t_logits, t_feats = net(tgt)
    loss = F.nll_loss(F.log_softmax(s_logits, dim=1), src.y)
    if disc == "JS":                                                         # :169-176
        preds = net.discriminator(grad_reverse(torch.cat([s_feats, t_feats], 0), alpha))
        # node mode: one label per node (:171-172); graph mode: len(batch) = its number of graphs (:173-174)
        ns, nt = (src.x.size(0), tgt.x.size(0)) if net.mode == "node" else (src.num_graphs, tgt.num_graphs)
        lab = torch.tensor([0] * ns + [1] * nt)
        dom = F.cross_entropy(preds, lab)
    elif disc == "MMD":                                                      # :177-182
        m = min(src.x.size(0), tgt.x.size(0)) if net.mode == "node" else min(src.num_graphs, tgt.num_graphs)
        dom = MMD(s_feats[:m], t_feats[:m], chunk_rows=mmd_chunk_rows, samples=mmd_samples)
    elif disc == "C":                                                        # :183-193
        # label-conditional: the discriminator sees [features, 8 x one-hot source label] / [features, 8 x softmax of
        # the target logits] (the softmax is differentiated: the target logits get a gradient through it)
        ratio = 8
        eye = torch.eye(net.num_classes)                                     # grade_base.py:183-202
        s_l_f = torch.cat([s_feats, ratio * eye[src.y]], dim=1)
        t_l_f = torch.cat([t_feats, ratio * F.softmax(t_logits, dim=1)], dim=1)
        preds = net.discriminator(grad_reverse(torch.cat([s_l_f, t_l_f], 0), alpha))
        ns, nt = (src.x.size(0), tgt.x.size(0)) if net.mode == "node" else (src.num_graphs, tgt.num_graphs)
        dom = F.cross_entropy(preds, torch.tensor([0] * ns + [1] * nt))
    else:
        raise NotImplementedError(disc)
    return loss + dom * weight, s_logits, t_logits


# ----------------------------------------------------------------------------
# the CPU baseline step (bench.py cpu_baseline, kind="port")
# ----------------------------------------------------------------------------
def a2gnn_train_step(net: A2GNNBase, opt: torch.optim.Optimizer, src: Graph, tgt: Graph,
                     alpha: float, s_pnums: int, t_pnums: int, adv: bool, weight: float,
                     mmd_chunk_rows: Optional[int] = None):
    """One iteration of the step loop a2gnn.py:308-319."""
    net.train()
    loss, s_logits, _ = a2gnn_forward_model(net, src, tgt, alpha, s_pnums, t_pnums, adv,
                                            weight, mmd_chunk_rows)
    val = loss.item()
    opt.zero_grad()
    loss.backward()
    opt.step()
    return val, s_logits


def neighbor_batches(g: Graph, num_hops: int, batch_size: int) -> List[Graph]:
    """The batches ``NeighborLoader(data, [-1] * num_hops, batch_size=batch_size)`` (a2gnn.py:260-277, ``shuffle=False``)
    yields.  PyG's published algorithm for fan-out -1 (no draw is made): seeds ``batch_size`` at a time in node order;
    hop l expands every node first reached in hop l-1 (hop 1: the seeds), in discovery order, and takes all of its
    in-edges in CSC order (edge list stably sorted by destination); nodes = seeds first, then discoveries in order;
    edges relabelled, grouped by destination in expansion order.  ``batch_size >= N``: the graph itself."""
    n = g.x.size(0)
    if batch_size >= n:
        return [g]
    ei = g.edge_index
    perm = torch.argsort(ei[1], stable=True)
    src = ei[0][perm].tolist()
    ptr = [0] + torch.cumsum(torch.bincount(ei[1], minlength=n), 0).tolist()
    out = []
    for start in range(0, n, batch_size):
        nodes = list(range(start, min(start + batch_size, n)))
        local = {v: i for i, v in enumerate(nodes)}
        frontier, rows, cols = list(nodes), [], []
        for _ in range(num_hops):
            reached = []
            for v in frontier:
                for u in src[ptr[v]:ptr[v + 1]]:
                    if u not in local:
                        local[u] = len(nodes)
                        nodes.append(u)
                        reached.append(u)
                    rows.append(local[u])
                    cols.append(local[v])
            frontier = reached
        n_id = torch.tensor(nodes, dtype=torch.long)
        out.append(Graph(g.x[n_id], torch.tensor([rows, cols], dtype=torch.long).reshape(2, -1), g.y[n_id]))
    return out


def a2gnn_fit(net: A2GNNBase, opt: torch.optim.Optimizer, src_batches: List[Graph], tgt_batches: List[Graph], epochs: int,
              s_pnums: int, t_pnums: int, adv: bool, weight: float):
    """The epoch loop of a2gnn.py:298-336 over given loaders' batches: per epoch alpha = 2 / (1 + e^{-10 p}) - 1 (:305-306),
    ``zip`` of the two loaders (:308: stops at the shorter one), one optimiser step per pair, ``epoch_loss`` = sum of
    ``loss.item()`` (:315), micro-F1 of the concatenated WHOLE-batch source logits (:321-330; single-label micro-F1 =
    accuracy).  Returns (losses, accs)."""
    losses, accs = [], []
    for epoch in range(epochs):
        alpha = 2. / (1. + math.exp(-10. * float(epoch) / epochs)) - 1
        tot, logits, labels = 0.0, [], []
        for sb, tb in zip(src_batches, tgt_batches):
            val, s_logits = a2gnn_train_step(net, opt, sb, tb, alpha, s_pnums, t_pnums, adv, weight)
            tot += val
            logits.append(s_logits.detach())
            labels.append(sb.y)
        losses.append(tot)
        accs.append(float((torch.cat(logits).argmax(1) == torch.cat(labels)).double().mean()))
    return losses, accs


def a2gnn_predict(net: A2GNNBase, batches: List[Graph], pnums: int):
    """``A2GNN.predict`` as written (a2gnn.py:384-411), its multi-batch behaviour included: for ``idx > 0`` the fresh
    ``logits`` OVERWRITES the accumulated one before it is concatenated with itself (:402-405 / :413-416), so with B > 1
    batches the result is the LAST batch's logits twice, beside the labels of ALL batches (whole batches: seeds and
    their sampled neighbours).  One batch: (logits, labels) of that batch."""
    net.eval()
    logits = labels = None
    with torch.no_grad():
        for idx, b in enumerate(batches):
            logits_b = net(b, pnums)
            if idx == 0:
                logits, labels = logits_b, b.y
            else:
                logits = torch.cat((logits_b, logits_b))
                labels = torch.cat((labels, b.y))
    return logits, labels


# ----------------------------------------------------------------------------
# PPMI graph construction, UDAGCN, AdaGCN  (a11, a13, a14)
# ----------------------------------------------------------------------------
# ------------------------------------------------------------------------- TDSS --
def two_hop_edges(edge_index: Tensor, num_nodes: int) -> Tensor:
    """TwoHopNeighbor.__call__ without attributes (tdss.py:67-79): the pattern of A.A (spspmm),
    self loops removed, concatenated with E and coalesced (sorted by (row, col), duplicates
    dropped).  Pure-Python sets: for the small graphs of the parity tests."""
    out = [set() for _ in range(num_nodes)]
    for r, c in zip(edge_index[0].tolist(), edge_index[1].tolist()):
        out[r].add(c)
    pairs = []
    for i in range(num_nodes):
        reach = set(out[i])
        for j in out[i]:
            reach |= {k for k in out[j] if k != i}
        pairs += [(i, k) for k in sorted(reach)]
    return torch.tensor(pairs, dtype=torch.long).t().reshape(2, -1)


def tdss_smoothness_khop(edge_index: Tensor, num_nodes: int, k: int) -> Tensor:
    """TDSS.smoothness, K-hop branch (tdss.py:374-386)."""
    ei = edge_index
    for _ in range(k - 1):
        ei = two_hop_edges(ei, num_nodes)
    if k == 1:                       # :375-376 passes no num_nodes: PyG infers max index + 1
        num_nodes = int(ei.max()) + 1 if ei.numel() else 0
    w = torch.ones(ei.size(1))
    ei, _ = add_remaining_self_loops(ei, w, 1.0, num_nodes)
    return ei


def laplacian_loss(features: Tensor, edge_index: Tensor) -> Tensor:
    """TDSS.compute_laplacian_loss (tdss.py:435-454)."""
    row, col = edge_index
    deg = torch.zeros(features.size(0)).index_add_(0, row, torch.ones(row.numel()))     # :441-442
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0                                                           # :443-444
    diff = features[row] * dis[row].view(-1, 1) - features[col] * dis[col].view(-1, 1)  # :446-448
    return diff.pow(2).sum(dim=1).sum() / 2.                                            # :450-454


def tdss_forward_model(net: A2GNNBase, src: Graph, tgt: Graph, smooth_ei: Tensor, s_pnums: int,
                       t_pnums: int, alpha: float, beta: float, mmd_samples=None):
    """tdss.py:241-312: CE(source) + alpha * MMD + beta * Laplacian(target features)."""
    source_logits = net(src, s_pnums)                                        # :275
    loss = F.nll_loss(F.log_softmax(source_logits, dim=1), src.y)
    sf = net.feat_bottleneck(src.x, src.edge_index, None, s_pnums)           # :286
    tf = net.feat_bottleneck(tgt.x, tgt.edge_index, None, t_pnums)           # :287
    loss = loss + alpha * MMD(sf, tf, samples=mmd_samples)                   # :301-303
    loss = loss + beta * laplacian_loss(tf, smooth_ei)                       # :306-307
    target_logits = net(tgt, t_pnums)                                        # :309
    return loss, source_logits, target_logits


def ppmi_raw_edges(edge_index: Tensor, path_len: int = 5, passes: int = 40):
    """ppmi_conv.py:56-169: ``passes`` (40 in the reference) rounds of random walks (length ~
    U{1..path_len}) from every node over the symmetrised neighbour sets, ``np.random`` stream,
    row-normalised visit counts -> PPMI = max(log(p / sum_a p * |targets| / path_len), 0) ->
    weighted edge list (float64).  Python containers are used exactly as in the reference
    because their iteration order is part of the result."""
    import numpy as np
    from collections import Counter
    adj = {}
    for a, b in edge_index.t().numpy():
        a, b = int(a), int(b)
        adj.setdefault(a, set()).add(b)
        adj.setdefault(b, set()).add(a)
    adj = {a: list(nb) for a, nb in adj.items()}
    walks = {}
    for _ in range(passes):                                                  # :119
        for a in adj:
            cur = a
            steps = np.random.randint(1, path_len + 1)                       # :122
            for _ in range(steps):
                nb = adj[cur]
                b = nb[np.random.randint(0, len(nb))]                        # :104-107
                walks.setdefault(a, Counter())[b] += 1
                cur = b
    normed = {}
    for a, c in walks.items():                                               # :109-117,150
        s = sum(c.values())
        normed[a] = {b: cnt / s for b, cnt in c.items()}
    prob_sums = Counter()
    for a, c in normed.items():                                              # :152-155
        for b, p in c.items():
            prob_sums[b] += p
    ei, ew = [], []
    for a, c in normed.items():                                              # :157-169
        for b, p in c.items():
            ei.append([a, b])
            ew.append(max(np.log(p / prob_sums[b] * len(prob_sums) / path_len), 0))
    return torch.tensor(ei).t(), torch.tensor(ew)                            # float64, like the reference


def ppmi_norm(edge_index: Tensor, num_nodes: int, path_len: int = 5, improved: bool = False):
    """ppmi_conv.py:171-184: the PPMI edge list + self loops + SOURCE-degree symmetric
    normalisation, evaluated in float64 and cast to float32 at the end."""
    edge_index, edge_weight = ppmi_raw_edges(edge_index, path_len)
    edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight,
                                                       2 if improved else 1, num_nodes)
    row, col = edge_index
    deg = torch.zeros(num_nodes, dtype=edge_weight.dtype).index_add_(0, row, edge_weight)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return edge_index, (dis[row] * edge_weight * dis[col]).type(torch.float32)  # :176-184


class PPMIConv(CachedGCNConv):
    """ppmi_conv.py:10-184: a CachedGCNConv whose cached graph is the PPMI graph."""

    def __init__(self, in_channels, out_channels, weight=None, bias=None, improved=False,
                 use_bias=True, path_len=5):
        super().__init__(in_channels, out_channels, weight, bias, improved, use_bias)
        self.path_len = path_len

    def forward(self, x, edge_index, cache_name="default_cache", edge_weight=None):
        x = torch.matmul(x, self.weight)
        if cache_name not in self.cache_dict:
            self.cache_dict[cache_name] = ppmi_norm(edge_index, x.size(0), self.path_len, self.improved)
        ei, norm = self.cache_dict[cache_name]
        out = propagate(ei, norm, x)
        return out + self.bias if self.bias is not None else out


class _UDAGNN(nn.Module):
    """udagcn_base.py:9-89: conv stack; the Dropout(0.1) modules sit in a plain list (never
    registered, never switched to eval) and the ctor's dropout argument is ignored."""

    def __init__(self, in_dim, hid_dim, gnn_type="gcn", num_layers=3, base_model=None, act=F.relu,
                 dropout_p=0.1, **kw):
        super().__init__()
        ws = [None] * num_layers if base_model is None else [c.weight for c in base_model.conv_layers]
        bs = [None] * num_layers if base_model is None else [c.bias for c in base_model.conv_layers]
        self.dropout_layers = [nn.Dropout(dropout_p) for _ in ws]
        self.act = act
        cls = PPMIConv if gnn_type == "ppmi" else CachedGCNConv
        dims = [in_dim] + [hid_dim] * num_layers
        self.conv_layers = nn.ModuleList(cls(dims[i], dims[i + 1], weight=ws[i], bias=bs[i], **kw)
                                         for i in range(num_layers))

    def forward(self, x, edge_index, cache_name):
        for i, conv in enumerate(self.conv_layers):
            x = conv(x, edge_index, cache_name)
            if i < len(self.conv_layers) - 1:
                x = self.dropout_layers[i](self.act(x))
        return x


class UDAGCNBase(nn.Module):
    """udagcn_base.py:92-267."""

    def __init__(self, in_dim, hid_dim, num_classes, num_layers=3, act=F.relu, ppmi=True, adv_dim=40,
                 dropout_p=0.1):
        super().__init__()
        self.ppmi = ppmi
        self.encoder = _UDAGNN(in_dim, hid_dim, "gcn", num_layers, act=act, dropout_p=dropout_p)
        if ppmi:
            self.ppmi_encoder = _UDAGNN(in_dim, hid_dim, "ppmi", num_layers, base_model=self.encoder,
                                        act=act, dropout_p=dropout_p, path_len=10)
        self.cls_model = nn.Sequential(nn.Linear(hid_dim, num_classes))
        self.domain_model = nn.Sequential(nn.Linear(hid_dim, adv_dim), nn.ReLU(), nn.Dropout(dropout_p),
                                          nn.Linear(adv_dim, 2))
        self.att_model = Attention(hid_dim)
        self.loss_func = nn.CrossEntropyLoss()

    def encode(self, data, cache_name):
        g = self.encoder(data.x, data.edge_index, cache_name)
        if not self.ppmi:
            return g
        p = self.ppmi_encoder(data.x, data.edge_index, cache_name)
        return self.att_model([g, p])


def udagcn_forward_model(net: UDAGCNBase, src: Graph, tgt: Graph, alpha: float, epoch: int,
                         epochs: int, mode: str = "node"):
    """udagcn.py:131-201.  ``mode='graph'`` (:168-170): embeddings mean-pooled per graph; the adjacency caches are
    emptied first -- the reference never invalidates them (cached_gcn_conv.py:132-136) and would aggregate every
    shuffled batch after the first over the first batch's edges (tests/golden/make_golden.py::fx_graph_trainers
    records the reference with the caches emptied the same way)."""
    if mode == "graph":
        for enc in (net.encoder, getattr(net, "ppmi_encoder", None)):
            for conv in (() if enc is None else enc.conv_layers):
                conv.cache_dict.clear()
    es, et = net.encode(src, "source"), net.encode(tgt, "target")
    if mode == "graph":
        es, et = global_mean_pool(es, src.batch), global_mean_pool(et, tgt.batch)
    s_logits = net.cls_model(es)
    loss = net.loss_func(s_logits, src.y)                                            # :172
    sd = net.domain_model(grad_reverse(es, alpha))
    td = net.domain_model(grad_reverse(et, alpha))
    loss_grl = net.loss_func(sd, torch.zeros(sd.size(0), dtype=torch.long)) \
        + net.loss_func(td, torch.ones(td.size(0), dtype=torch.long))               # :177-189
    loss = loss + loss_grl                                                           # :190 (cls + (src + tgt))
    t_logits = net.cls_model(et)
    p = torch.clamp(F.softmax(t_logits, dim=-1), min=1e-9, max=1.0)
    ent = torch.mean(torch.sum(-p * torch.log(p), dim=-1))                           # :193-197
    return loss + ent * (epoch / epochs * 0.01), s_logits, t_logits                  # :199


# ------------------------------------------------------------------------ DGSDA --
def sym_laplacian(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int):
    """PyG ``get_laplacian(..., normalization='sym')`` as called at dgsda_base.py:128: self loops
    removed, degree over ``row``, off-diagonal ``-d^-1/2 w d^-1/2`` followed by N diagonal ones."""
    row, col = edge_index
    keep = row != col
    row, col = row[keep], col[keep]
    w = torch.ones(row.numel()) if edge_weight is None else edge_weight[keep]
    deg = torch.zeros(num_nodes).index_add_(0, row, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float('inf'), 0)
    w = dis[row] * w * dis[col]
    loops = torch.arange(num_nodes)
    return torch.stack([torch.cat([row, loops]), torch.cat([col, loops])]), torch.cat([-w, torch.ones(num_nodes)])


class BernProp(nn.Module):
    """dgsda_base.py:11-183, the reference's evaluation order: tmp[j] = (2I-L)^j x, then for every i
    the chain L^(i+1) tmp[K-i-1]."""

    def __init__(self, K: int, is_source_domain: bool = True):
        super().__init__()
        self.K = K
        self.temp = nn.Parameter(torch.ones(K + 1) if is_source_domain else torch.linspace(1, 0, K + 1),
                                 requires_grad=is_source_domain)

    def forward(self, x, edge_index, edge_weight=None):
        from math import comb
        K, n = self.K, x.size(0)
        TEMP = F.relu(self.temp)
        ei1, norm1 = sym_laplacian(edge_index, edge_weight, n)                          # :128
        loops = torch.arange(n)
        ei2 = torch.cat([ei1, torch.stack([loops, loops])], dim=1)                      # :130 add_self_loops
        norm2 = torch.cat([-norm1, torch.full((n,), 2.0)])
        tmp = [x]
        for _ in range(K):
            x = propagate(ei2, norm2, x)
            tmp.append(x)
        out = (comb(K, 0) / (2 ** K)) * TEMP[0] * tmp[K]
        for i in range(K):
            x = tmp[K - i - 1]
            x = propagate(ei1, norm1, x)
            for _ in range(i):
                x = propagate(ei1, norm1, x)
            out = out + (comb(K, i + 1) / (2 ** K)) * TEMP[i + 1] * x
        return out


class DGSDABase(nn.Module):
    """dgsda_base.py:186-315."""

    def __init__(self, features, hidden, classes, dprate=0.0, K=15):
        super().__init__()
        self.lin1 = nn.Linear(features, hidden)
        self.lin2 = nn.Linear(hidden, classes)
        self.prop1, self.prop2, self.prop3 = BernProp(K), BernProp(K), BernProp(K)
        self.dprate = dprate

    def get_props(self, x, edge_index, is_source_domain=True):
        x = F.dropout(x, p=self.dprate, training=self.training)
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=self.dprate, training=self.training)
        return self.prop1(x, edge_index) if is_source_domain else self.prop2(x, edge_index)

    def forward(self, data, is_source_domain=True):
        x = self.get_props(data.x, data.edge_index, is_source_domain)
        x = F.dropout(x, p=self.dprate, training=self.training)
        x = self.lin2(x)
        x = F.dropout(x, p=self.dprate, training=self.training)
        return self.prop3(x, data.edge_index)


def dgsda_entropy(output: Tensor) -> Tensor:
    """DGSDA.entropy_minimization_loss (dgsda.py:222-227)."""
    probs, log_probs = F.softmax(output, dim=1), F.log_softmax(output, dim=1)
    a = torch.sum(probs, dim=0)
    return -torch.sum(probs * log_probs / (a / torch.sum(a)), dim=1).mean()


def dgsda_forward_model(net: DGSDABase, src: Graph, tgt: Graph, alpha: float, beta: float, gamma: float,
                        mmd_samples=None):
    """dgsda.py:144-196."""
    s_logits = net(src)
    loss = F.nll_loss(F.log_softmax(s_logits, dim=1), src.y)
    loss = loss + F.l1_loss(net.prop1.temp, net.prop2.temp) * alpha
    sf, tf = F.relu(net.lin1(src.x)), F.relu(net.lin1(tgt.x))
    loss = loss + MMD(sf, tf, samples=mmd_samples) * beta
    loss = loss + dgsda_entropy(net(tgt, False)) * gamma
    return loss, s_logits


# ----------------------------------------------------------------------- StruRW --
def _mean_aggregate_t2s(edge_index: Tensor, msg: Tensor, n: int) -> Tensor:
    """PyG aggr='mean' with flow='target_to_source': messages averaged at edge_index[0] over their
    NUMBER (not their weight); nodes without messages stay zero."""
    out = torch.zeros(n, msg.size(1)).index_add_(0, edge_index[0], msg)
    cnt = torch.zeros(n).index_add_(0, edge_index[0], torch.ones(edge_index.size(1)))
    return out / cnt.clamp(min=1).view(-1, 1)


class GSReweight(nn.Module):
    """GS_reweight (reweight_gnn.py:252-372): lin on the gathered neighbour rows, message
    (1-lmda) m + lmda w_e m, mean aggregation, agg_lin(cat(aggr, x)), relu."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels)
        self.agg_lin = nn.Linear(out_channels + in_channels, out_channels)

    def forward(self, x, edge_index, edge_weight, lmda):
        m = self.lin(x.index_select(0, edge_index[1]))                                   # :308-309
        m = (1 - lmda) * m + lmda * (edge_weight.view(-1, 1) * m)                        # :310
        aggr = _mean_aggregate_t2s(edge_index, m, x.size(0))
        return F.relu(self.agg_lin(torch.cat((aggr, x), dim=-1)))                        # :342-347


class GCNReweight(nn.Module):
    """GCN_reweight (reweight_gnn.py:51-250) as ReweightGNN builds it (aggr = pooling): with
    aggr='mean' the edge list is gcn-normalised WITHOUT self loops (column degree, unit weights)
    and the messages are additionally averaged; with aggr='add' no normalisation at all."""

    def __init__(self, in_channels, out_channels, aggr="mean"):
        super().__init__()
        self.aggr = aggr
        self.lin = _Lin(in_channels, out_channels)
        glorot_(self.lin.weight)                       # __init__ draws, reset_parameters() draws again (:115-116)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index, edge_weight, lmda):
        rw, n = edge_weight, x.size(0)
        w = torch.ones_like(rw)
        if self.aggr != "add":                                                           # :96-99, :151-160
            deg = torch.zeros(n).index_add_(0, edge_index[1], w)
            dis = deg.pow(-0.5)
            dis.masked_fill_(dis == float("inf"), 0)
            w = dis[edge_index[0]] * w * dis[edge_index[1]]
        h = x @ self.lin.weight.t()
        m = w.view(-1, 1) * h.index_select(0, edge_index[1])                             # :222
        m = (1 - lmda) * m + lmda * (rw.view(-1, 1) * m)                                 # :223
        if self.aggr == "add":
            out = torch.zeros(n, m.size(1)).index_add_(0, edge_index[0], m)
        else:
            out = _mean_aggregate_t2s(edge_index, m, n)
        return out + self.bias


class ReweightGNN(nn.Module):
    """reweight_gnn.py:375-502.  ``conv`` lists prop_input once and the SAME prop_hidden module
    gnn_layers-1 times; dropout is applied with training=True regardless of the mode (:488)."""

    def __init__(self, input_dim, gnn_dim, output_dim, cls_dim, gnn_layers=3, cls_layers=2, backbone="GS",
                 pooling="mean", dropout=0.5, bn=False, rw_lmda=1.0):
        super().__init__()
        if backbone == "GCN":
            self.prop_input, self.prop_hidden = GCNReweight(input_dim, gnn_dim, pooling), GCNReweight(gnn_dim, gnn_dim, pooling)
        else:
            self.prop_input, self.prop_hidden = GSReweight(input_dim, gnn_dim), GSReweight(gnn_dim, gnn_dim)
        self.dropout, self.bn, self.lmda = dropout, bn, rw_lmda
        self.conv = nn.ModuleList([self.prop_input] + [self.prop_hidden] * (gnn_layers - 1))
        self.bns = nn.ModuleList(nn.BatchNorm1d(gnn_dim) for _ in range(gnn_layers - 1))
        self.bn_mlp = nn.BatchNorm1d(cls_dim)
        dims = [gnn_dim, output_dim] if cls_layers == 1 else [gnn_dim] + [cls_dim] * (cls_layers - 1) + [output_dim]
        self.mlp_classify = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, data, h):
        x = h
        for layer in self.conv:
            x = F.relu(layer(x, data.edge_index, data.edge_weight, self.lmda))
            x = F.dropout(x, p=self.dropout)
        y = x
        for i, lin in enumerate(self.mlp_classify):
            y = lin(y)
            if i != len(self.mlp_classify) - 1:
                if self.bn:
                    y = self.bn_mlp(y)
                y = F.relu(y)
        return x, y


def strurw_edge_weights(src: Graph, tgt: Graph, tgt_pred: Tensor, num_classes: int) -> Tensor:
    """StruRW.cal_reweight + cal_edge_prob_sep (strurw.py:446-547): class-pair edge probabilities of
    the source (labels) and the target (pseudo labels), ratio tgt/src with inf/nan -> 1, and for a
    source edge (u, v): weight = ratio[label(v), label(u)] (:476-481: edge_index[0] against the
    column class j, edge_index[1] against the row class i).  Counting edges per class pair replaces
    the dense Y^T A Y products (to_dense_adj sums duplicate edges, so the counts agree)."""
    def pair_prob(ei, lab, n, eps):
        c = num_classes
        cnt = torch.zeros(c * c, dtype=torch.float64).index_add_(
            0, lab[ei[0]] * c + lab[ei[1]], torch.ones(ei.size(1), dtype=torch.float64)).view(c, c)
        per = torch.bincount(lab, minlength=c).double()
        return cnt / (per.view(-1, 1) * per.view(1, -1) + eps)
    src_prob = pair_prob(src.edge_index, src.y, src.x.size(0), 0.0)                    # :540
    tgt_prob = pair_prob(tgt.edge_index, tgt_pred, tgt.x.size(0), 1e-12)               # :541
    ratio = tgt_prob / src_prob
    ratio[torch.isinf(ratio)] = 1
    ratio[torch.isnan(ratio)] = 1
    lab = src.y
    return ratio[lab[src.edge_index[1]], lab[src.edge_index[0]]].float()


def strurw_forward_model(net: ReweightGNN, src: Graph, tgt: Graph, alpha: float, epoch: int, mode="erm",
                         reweight=True, pseudo=True, ew_start=100, ew_freq=20, disc: Optional[nn.Module] = None,
                         num_classes: int = 0, mmd_samples=None):
    """strurw.py:189-257.  ``src.edge_weight`` is replaced in place when the re-weighting fires."""
    t_feat, t_logits = net(tgt, tgt.x)
    t_pred = F.softmax(t_logits, dim=1).max(dim=1)[1]
    if reweight and (epoch + 1) >= ew_start:
        if (pseudo and (epoch + 1) % ew_freq == 0) or (not pseudo and epoch == ew_start - 1):
            src.edge_weight = strurw_edge_weights(src, tgt, t_pred, num_classes)
    s_feat, s_logits = net(src, src.x)
    loss = F.nll_loss(F.log_softmax(s_logits, dim=1), src.y)
    if mode == "adv":
        sd, td = disc(grad_reverse(s_feat, alpha)), disc(grad_reverse(t_feat, alpha))
        lab = torch.tensor([0] * src.x.shape[0] + [1] * tgt.x.shape[0])
        loss = loss + F.cross_entropy(torch.cat([sd, td], 0), lab)
    elif mode == "mmd":
        loss = loss + MMD(s_feat, t_feat, samples=mmd_samples)
    return loss, s_logits, t_logits


class MixUpGCNConv(nn.Module):
    """MixUpGCNConv (mixup_gcnconv.py:69-247), flow source_to_target, add aggregation: the edge list is
    gcn-normalised with UNIT weights and no self loops (:204-214; the weight it is handed is the
    re-weighting factor, set aside as ``edge_rw`` :200-201), the message is
    ``(1-lmda) n_e x_j + lmda rw_e n_e x_j`` (:242-245) and the centre term ``lin_cen(x_cen)`` plus the
    bias is added to the aggregate (:231-236).  RNG order of __init__: lin, lin_cen, lin again (:122-133)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = _Lin(in_channels, out_channels)
        self.lin_cen = _Lin(in_channels, out_channels)
        glorot_(self.lin.weight)                        # reset_parameters(): lin only
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, x_cen, edge_index, edge_weight, lmda=1):
        n = x.size(0)
        deg = torch.zeros(n).index_add_(0, edge_index[1], torch.ones(edge_index.size(1)))
        dis = deg.pow(-0.5)
        dis.masked_fill_(dis == float("inf"), 0)
        norm = dis[edge_index[0]] * dis[edge_index[1]]
        h = x @ self.lin.weight.t()
        m = norm.view(-1, 1) * h.index_select(0, edge_index[0])
        m = (1 - lmda) * m + lmda * (edge_weight.view(-1, 1) * m)
        out = torch.zeros(n, m.size(1)).index_add_(0, edge_index[1], m)
        return out + x_cen @ self.lin_cen.weight.t() + self.bias


class MixupBase(nn.Module):
    """mixup_base.py:10-200: per layer three convolutions -- the plain one (x, x), the mixed-centre one
    on the graph (x, x_mix) and the mixed-centre one on the SHUFFLED graph (x[perm], x_mix, edge_index_b)
    -- and ``x_mix <- dropout(lam * act(new) + (1-lam) * act(new_b))``.  Needs num_layers >= 2 (:150)."""

    def __init__(self, in_dim, hid_dim, num_classes, num_layers=1, dropout=0.1, act=F.relu, rw_lmda=0.8):
        super().__init__()
        self.dropout, self.act, self.rw_lmda = dropout, act, rw_lmda
        self.convs = nn.ModuleList([MixUpGCNConv(in_dim, hid_dim)] +
                                   [MixUpGCNConv(hid_dim, hid_dim) for _ in range(num_layers - 1)])
        self.cls = nn.Linear(hid_dim, num_classes)

    def _drop(self, x):
        return F.dropout(x, p=self.dropout, training=self.training)

    def feat_bottleneck(self, x, edge_index, edge_index_b, lam, id_new_value_old, edge_weight):
        c, rw = self.convs, self.rw_lmda
        x1 = self._drop(self.act(c[0](x, x, edge_index, edge_weight, rw)))              # :146-148
        x2 = self._drop(self.act(c[1](x1, x1, edge_index, edge_weight, rw)))            # :150-152
        x0_b, x1_b = x[id_new_value_old], x1[id_new_value_old]
        x_mix = x * lam + x0_b * (1 - lam)                                              # :157
        new_x1 = self.act(c[0](x, x_mix, edge_index, edge_weight, rw))
        new_x1_b = self.act(c[0](x0_b, x_mix, edge_index_b, edge_weight, rw))
        x1_mix = self._drop(new_x1 * lam + new_x1_b * (1 - lam))                        # :165-166
        new_x2 = self.act(c[1](x1, x1_mix, edge_index, edge_weight, rw))
        new_x2_b = self.act(c[1](x1_b, x1_mix, edge_index_b, edge_weight, rw))
        x2_mix = self._drop(new_x2 * lam + new_x2_b * (1 - lam))                        # :174-175
        x, x_mix = x2, x2_mix
        for i in range(2, len(c)):                                                      # :180-194
            x_t = self._drop(self.act(c[i](x, x, edge_index, edge_weight, rw)))
            x_b = x[id_new_value_old]
            new_x = self.act(c[i](x, x_mix, edge_index, edge_weight, rw))
            new_x_b = self.act(c[i](x_b, x_mix, edge_index_b, edge_weight, rw))
            x_mix = self._drop(new_x * lam + new_x_b * (1 - lam))
            x = x_t
        return x_mix

    def forward(self, x, edge_index, edge_index_b, lam, id_new_value_old, edge_weight):
        return self.cls(self.feat_bottleneck(x, edge_index, edge_index_b, lam, id_new_value_old, edge_weight))


def strurw_shuffle(edge_index: Tensor, perm) -> Tensor:
    """StruRW.shuffle_data + id_node (strurw.py:702-758): ``perm`` = id_new_value_old (new position -> old
    node); the edge list relabelled old -> new, edge order unchanged."""
    perm = torch.as_tensor(perm, dtype=torch.long)
    old_to_new = torch.zeros(perm.numel(), dtype=torch.long)
    old_to_new[perm] = torch.arange(perm.numel())
    return torch.stack([old_to_new[edge_index[0]], old_to_new[edge_index[1]]], dim=0)


def strurw_forward_model_mixup(net: MixupBase, src: Graph, tgt: Graph, epoch: int, lam: float, perm,
                               reweight=True, pseudo=True, ew_start=100, ew_freq=20, num_classes: int = 0):
    """strurw.py:259-313; the caller draws ``lam`` (np.random.beta(4, 4)) and ``perm`` (np.random.shuffle of
    arange) in that order, as :292-293 do."""
    import numpy as _np
    n_t = tgt.x.size(0)
    t_logits = net(tgt.x, tgt.edge_index, tgt.edge_index, 1, _np.arange(n_t), tgt.edge_weight)
    t_pred = F.softmax(t_logits, dim=1).max(dim=1)[1]
    if reweight and (epoch + 1) >= ew_start:
        if (pseudo and (epoch + 1) % ew_freq == 0) or (not pseudo and epoch == ew_start - 1):
            src.edge_weight = strurw_edge_weights(src, tgt, t_pred, num_classes)
    s_logits = net(src.x, src.edge_index, strurw_shuffle(src.edge_index, perm), lam, perm, src.edge_weight)
    loss = F.nll_loss(F.log_softmax(s_logits, dim=1), src.y)
    return loss, s_logits, t_logits


# ---------------------------------------------------------------------- SpecReg --
def specreg_gradient_penalty(critic: nn.Module, x_src: Tensor, x_tgt: Tensor) -> Tensor:
    """SpecReg.calculate_gradient_penalty (specreg.py:380-419): no interpolation -- the critic's
    input gradient at the source and target encodings themselves."""
    x = torch.cat([x_src, x_tgt], dim=0).requires_grad_(True)
    out = critic(x)
    grad = torch.autograd.grad(outputs=out, inputs=x, grad_outputs=torch.ones(out.shape),
                               create_graph=True, retain_graph=True, only_inputs=True)[0]
    grad = grad.view(grad.shape[0], -1)
    return torch.mean((grad.norm(2, dim=1) - 1) ** 2)


def specreg_forward_model(net: UDAGCNBase, critic: nn.Module, c_opt, src: Graph, tgt: Graph,
                          eivec_s: Optional[Tensor], eivec_t: Optional[Tensor], epoch: int, epochs: int,
                          reg_mode=True, gamma_adv=0.1, thr_smooth=-1., gamma_smooth=0.01,
                          thr_mfr=-1., gamma_mfr=0.01):
    """specreg.py:150-226: source CE + gamma_adv * Wasserstein gap (critic trained 5 steps with
    gradient penalty) + spectral smoothness / maximum-frequency-response hinges on the encodings
    projected onto the Laplacian eigenvectors + annealed target entropy."""
    es, et = net.encode(src, "source"), net.encode(tgt, "target")
    s_logits = net.cls_model(es)
    loss = net.loss_func(s_logits, src.y)                                            # :185
    xs, xt = es.detach(), et.detach()
    for _ in range(5):                                                               # :188-194
        c_opt.zero_grad()
        gap = critic(xs).mean() - critic(xt).mean()
        adv = -gap + 10 * specreg_gradient_penalty(critic, xs, xt)
        adv.backward()
        c_opt.step()
    loss = loss + (critic(es).mean() - critic(et).mean()) * gamma_adv                # :196-197
    if reg_mode:                                                                     # :199-209
        fs = torch.einsum('nm,md->nd', eivec_s, es)
        ft = torch.einsum('nm,md->nd', eivec_t, et)
        if thr_smooth > 0:
            ds, dt = (fs[:-1] - fs[1:]).abs(), (ft[:-1] - ft[1:]).abs()
            loss = loss + (F.relu(ds - thr_smooth).mean() + F.relu(dt - thr_smooth).mean()) * gamma_smooth
        if thr_mfr > 0:
            loss = loss + (F.relu(fs.abs() - thr_mfr).mean() + F.relu(ft.abs() - thr_mfr).mean()) * gamma_mfr
    t_logits = net.cls_model(et)
    p = torch.clamp(F.softmax(t_logits, dim=-1), min=1e-9, max=1.0)
    ent = torch.mean(torch.sum(-p * torch.log(p), dim=-1))                           # :212-216
    return loss + ent * (epoch / epochs * 0.01), s_logits, t_logits


def specreg_critic(hid_dim: int) -> nn.Module:
    """specreg.py:281-287."""
    return nn.Sequential(nn.Linear(hid_dim, hid_dim), nn.ReLU(), nn.Linear(hid_dim, hid_dim), nn.ReLU(),
                         nn.Linear(hid_dim, 1))


class _AdaGNN(nn.Module):
    """adagcn_base.py:11-97 (gnn_type='gcn'): L GCNConv, act + Dropout between layers."""

    def __init__(self, in_dim, hid_dim, num_layers, act, dropout_p):
        super().__init__()
        self.act = act
        dims = [in_dim] + [hid_dim] * num_layers
        self.conv_layers = nn.ModuleList(GCNConv(dims[i], dims[i + 1]) for i in range(num_layers))
        self.dropout = nn.Dropout(dropout_p)

    def forward(self, x, edge_index):
        for i, conv in enumerate(self.conv_layers):
            x = conv(x, edge_index)
            if i < len(self.conv_layers) - 1:
                x = self.dropout(self.act(x))
        return x


class AdaGCNBase(nn.Module):
    """adagcn_base.py:99-181 (node mode).  The ctor's ``dropout`` never reaches the stack
    (:145 builds GNN without it, so its default 0.1 at :39 always applies)."""

    def __init__(self, in_dim, hid_dim, num_classes, num_layers=3, act=F.relu, dropout_p=0.1, mode="node"):
        super().__init__()
        self.encoder = _AdaGNN(in_dim, hid_dim, num_layers, act, dropout_p)
        self.cls_model = nn.Sequential(nn.Linear(hid_dim, num_classes))
        self.loss_func = nn.CrossEntropyLoss()
        self.mode = mode

    def forward(self, data):
        x = self.encoder(data.x, data.edge_index)
        return global_mean_pool(x, data.batch) if self.mode == "graph" else x      # adagcn_base.py:93-94


def adagcn_gradient_penalty(disc: nn.Module, es: Tensor, et: Tensor) -> Tensor:
    """adagcn.py:387-454: WGAN-GP over cat(source, target, interpolates); the interpolation
    weights come from the CPU generator (``torch.rand`` then ``.to(device)``)."""
    ns, nt = es.shape[0], et.shape[0]
    if ns < nt:
        hs = torch.cat((es, es), 0)
        ht = torch.cat((et[0:ns], et[-ns:]), 0)
        a = torch.rand((2 * ns, 1))
    elif ns > nt:
        hs = torch.cat((es[0:nt], es[-nt:]), 0)
        ht = torch.cat((et, et), 0)
        a = torch.rand((2 * nt, 1))
    else:
        hs, ht = es, et
        a = torch.rand((nt, 1))
    inter = ht + a * (hs - ht)
    inputs = torch.cat((es, et, inter), 0)
    scores = disc(inputs)
    grad = torch.autograd.grad(inputs=inputs, outputs=scores, grad_outputs=torch.ones_like(scores),
                               create_graph=True, retain_graph=True, only_inputs=True)[0]
    return torch.mean((grad.view(grad.shape[0], -1).norm(2, dim=1) - 1) ** 2)


def adagcn_forward_model(net: AdaGCNBase, disc: nn.Module, c_opt, src: Graph, tgt: Graph,
                         gp_weight: float, domain_weight: float, critic_steps: int = 10):
    """adagcn.py:138-198: ``critic_steps`` Wasserstein-critic updates, then the encoder loss."""
    for _ in range(critic_steps):
        es, et = net(src), net(tgt)
        gp = adagcn_gradient_penalty(disc, es, et)
        dis = -torch.abs(torch.mean(disc(es).reshape(-1)) - torch.mean(disc(et).reshape(-1)))
        c_opt.zero_grad()
        (dis + gp_weight * gp).backward()
        c_opt.step()
    es, et = net(src), net(tgt)
    s_logits = net.cls_model(es)
    loss = net.loss_func(s_logits, src.y)
    dis = torch.abs(torch.mean(disc(es).reshape(-1)) - torch.mean(disc(et).reshape(-1)))
    return loss + dis * domain_weight, s_logits, net.cls_model(et)


# ----------------------------------------------------------------------------
# GNNBase backbones, GNN and DANE trainers  (a15)
# ----------------------------------------------------------------------------
def _sum_aggregate(x: Tensor, edge_index: Tensor) -> Tensor:
    return torch.zeros_like(x).index_add_(0, edge_index[1], x.index_select(0, edge_index[0]))


def _pyg_linear_reset(lin: nn.Linear) -> None:
    """PyG ``Linear.reset_parameters()`` with its default initialisers (dense/linear.py: ``kaiming_uniform(weight,
    fan=in, a=sqrt(5))`` = U(-b, b), b = sqrt(6 / ((1 + a^2) in)); ``inits.uniform(in, bias)`` = U(+-1/sqrt(in))):
    weight first, then bias -- the order the init RNG stream is consumed in."""
    fan = lin.weight.size(1)
    bound = math.sqrt(6 / ((1 + math.sqrt(5) ** 2) * fan))
    with torch.no_grad():
        lin.weight.uniform_(-bound, bound)
        if lin.bias is not None:
            b = 1.0 / math.sqrt(fan)
            lin.bias.uniform_(-b, b)


class SAGEConv(nn.Module):
    """PyG ``SAGEConv`` defaults as gnn_base.py:73-79 uses it (aggr='mean', root_weight=True):
    ``lin_l(mean_{j->i} x_j) + lin_r(x_i)``; no self loops; lin_l carries the bias.  Init: each PyG ``Linear`` draws
    in its constructor and again in ``SAGEConv.reset_parameters()`` (lin_l.W, lin_l.b, lin_r.W, twice over).
    Pinned since round 5 by ``tests/golden/gnn_fit2_sage.npz``: the reference's gnn_base.py / gnn.py executed on the
    stub's SAGEConv (tests/golden/_pyg_stub.py, assumption 13b)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        # torch's constructor draws stand where PyG's ``Linear.__init__`` draws (same counts, weight then bias) ...
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)
        _pyg_linear_reset(self.lin_l)                  # ... and SAGEConv.reset_parameters() sets the values
        _pyg_linear_reset(self.lin_r)

    def forward(self, x, edge_index, edge_weight=None):
        cnt = torch.zeros(x.size(0), dtype=x.dtype).index_add_(
            0, edge_index[1], torch.ones(edge_index.size(1), dtype=x.dtype))
        mean = _sum_aggregate(x, edge_index) / cnt.clamp(min=1).unsqueeze(1)
        return self.lin_l(mean) + self.lin_r(x)


class GINConv(nn.Module):
    """PyG ``GINConv(nn, train_eps=True)`` (gnn_base.py:89-95): ``nn(sum_j x_j + (1+eps) x_i)``, eps initialised to 0;
    its ``reset_parameters()`` re-draws every child of ``nn`` (the torch ``Linear`` of gnn_base.py:89 is initialised
    twice).  Pinned since round 5 by ``tests/golden/gnn_fit2_gin.npz`` (stub assumption 13c)."""

    def __init__(self, net: nn.Module):
        super().__init__()
        self.nn = net
        self.eps = nn.Parameter(torch.zeros(1))
        for child in net.children():
            child.reset_parameters()

    def forward(self, x, edge_index, edge_weight=None):
        return self.nn(_sum_aggregate(x, edge_index) + (1 + self.eps) * x)


class GATConv(nn.Module):
    """PyG ``GATConv(heads=1, concat=False)`` (gnn_base.py:81-87): x' = W x; self loops
    (existing removed, one added per node); e_ij = LeakyReLU_0.2(a_src.x'_j + a_dst.x'_i);
    alpha = softmax over the incoming edges of i; out_i = sum_j alpha_ij x'_j + bias.  Init: the shared ``lin`` is
    glorot-drawn by its constructor and again by ``reset_parameters()``, which then draws att_src and att_dst.
    Pinned since round 5 by ``tests/golden/gnn_fit2_gat.npz`` (stub assumption 13d)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = _Lin(in_channels, out_channels)
        self.att_src = nn.Parameter(torch.empty(1, 1, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, 1, out_channels))
        glorot_(self.lin.weight)                       # reset_parameters(): lin again, then the attention vectors
        glorot_(self.att_src); glorot_(self.att_dst)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index, edge_attr=None):
        n = x.size(0)
        keep = edge_index[0] != edge_index[1]
        loops = torch.arange(n)
        ei = torch.cat([edge_index[:, keep], torch.stack([loops, loops])], dim=1)
        h = F.linear(x, self.lin.weight)
        a_s = (h * self.att_src.view(1, -1)).sum(-1)
        a_d = (h * self.att_dst.view(1, -1)).sum(-1)
        e = F.leaky_relu(a_s[ei[0]] + a_d[ei[1]], 0.2)
        m = torch.full((n,), float("-inf")).scatter_reduce(0, ei[1], e.detach(), reduce="amax")
        p = torch.exp(e - m[ei[1]])
        z = torch.zeros(n).index_add_(0, ei[1], p) + 1e-16
        alpha = p / z[ei[1]]
        out = torch.zeros_like(h).index_add_(0, ei[1], alpha.unsqueeze(1) * h[ei[0]])
        return out + self.bias


class GNNBase(nn.Module):
    """gnn_base.py:11-203 (``mode='graph'``: mean readout per graph :133-134, linear classifier :97-98, :200-203)."""

    def __init__(self, in_dim, hid_dim, num_classes, num_layers=1, dropout=0.1, act=F.relu, gnn="gcn", mode="node"):
        super().__init__()
        self.dropout, self.act, self.mode = dropout, act, mode
        mk = {"gcn": GCNConv, "sage": SAGEConv, "gat": GATConv,
              "gin": lambda a, b: GINConv(nn.Sequential(nn.Linear(a, b)))}[gnn]
        dims = [in_dim] + [hid_dim] * num_layers
        self.convs = nn.ModuleList(mk(dims[i], dims[i + 1]) for i in range(num_layers))
        self.cls = mk(hid_dim, num_classes) if mode == "node" else nn.Linear(hid_dim, num_classes)

    def feat_bottleneck(self, x, edge_index, edge_weight=None):                 # :139-171
        for i, conv in enumerate(self.convs):
            x = conv(x, edge_index, edge_weight)
            if i < len(self.convs) - 1:
                x = F.dropout(self.act(x), p=self.dropout, training=self.training)
        return x

    def feat_classifier(self, x, edge_index, edge_weight=None):                 # :173-203
        return self.cls(x, edge_index, edge_weight) if self.mode == "node" else self.cls(x)

    def forward(self, x, edge_index, edge_weight=None, batch=None):              # :97-137
        x = self.feat_bottleneck(x, edge_index, edge_weight)
        if self.mode == "graph":
            x = global_mean_pool(x, batch)
        return F.log_softmax(self.feat_classifier(x, edge_index, edge_weight), dim=1)


def dane_l_gcn(embedding, nodes_weight, idx_u, idx_v, k, sample_size):
    """dane.py:357-389: skip-gram edge loss with degree^0.75 negative sampling.  (The negative
    indices address POSITIONS of the unique-source list, as the reference does.)"""
    eu, ev = embedding[idx_u], embedding[idx_v]
    neg = [embedding[torch.multinomial(nodes_weight, sample_size, replacement=False)] for _ in range(k)]
    loss = -torch.sum(F.logsigmoid(torch.sum(eu * ev, dim=1)))
    for i in range(k):
        loss = loss - torch.sum(F.logsigmoid(torch.sum(eu * neg[i] * (-1), dim=1)))
    return loss


def dane_forward_model(gnn: GNNBase, disc: nn.Module, g_opt, d_opt, src: Graph, tgt: Graph, k: int,
                       sample_size: int):
    """dane.py:145-180 + train_d :301-355 + train_g :426-516 (train_mode='unsup').  Graph mode (``gnn.mode``): the
    embeddings are mean-pooled per graph (:323-331, :448-456) and the skip-gram term is absent (:492-493)."""
    graph = gnn.mode == "graph"

    def embed(d):
        e = gnn.feat_bottleneck(d.x, d.edge_index)
        return global_mean_pool(e, d.batch) if graph else e

    d_loss = 0.0
    for _ in range(5):
        gnn.eval()
        es, et = embed(src), embed(tgt)
        i_s = torch.multinomial(torch.ones(es.shape[0]), 8 * sample_size, replacement=True)
        i_t = torch.multinomial(torch.ones(et.shape[0]), 8 * sample_size, replacement=True)
        d_opt.zero_grad()
        loss = (disc(es[i_s]) ** 2).mean() + ((disc(et[i_t]) - 1) ** 2).mean()
        loss.backward()
        d_opt.step()
        d_loss = loss.item()
    gnn.train()
    es = embed(src)
    out_s = gnn.feat_classifier(es, src.edge_index)
    et = embed(tgt)
    i_s = torch.multinomial(torch.ones(es.shape[0]), 8 * sample_size, replacement=True)
    i_t = torch.multinomial(torch.ones(et.shape[0]), 8 * sample_size, replacement=True)
    l_adv = (disc(et[i_t]) ** 2).mean() + ((disc(es[i_s]) - 1) ** 2).mean()
    if graph:
        l_gcn = 0
    else:
        e_s = torch.multinomial(torch.ones(src.edge_index.shape[1]), sample_size, replacement=False)
        e_t = torch.multinomial(torch.ones(tgt.edge_index.shape[1]), sample_size, replacement=False)
        w_s = torch.pow(torch.unique(src.edge_index[0], return_counts=True)[1], 0.75)
        w_t = torch.pow(torch.unique(tgt.edge_index[0], return_counts=True)[1], 0.75)
        l_gcn = dane_l_gcn(es, w_s, src.edge_index[0][e_s], src.edge_index[1][e_s], k, sample_size) + \
            dane_l_gcn(et, w_t, tgt.edge_index[0][e_t], tgt.edge_index[1][e_t], k, sample_size)
    loss = l_gcn + F.cross_entropy(out_s, src.y) + l_adv * 0.1
    g_opt.zero_grad()
    loss.backward()
    g_opt.step()
    kw_s, kw_t = (dict(batch=src.batch), dict(batch=tgt.batch)) if graph else ({}, {})
    return d_loss + loss.item(), gnn(src.x, src.edge_index, **kw_s), gnn(tgt.x, tgt.edge_index, **kw_t)
