"""Build-authored, test-only stand-in for the third-party modules the reference's
hot-path files import (torch_geometric, torch_scatter, torch_sparse), used ONLY by
``make_golden.py`` in the build container to execute the reference's own files
(loaded by path from the reference checkout) and record golden vectors.

None of those wheels is vendored in the reference or installable here, and the
reference has no tests pinning results at this boundary, so the semantics below are
restated from PyG's published behaviour (>=2.4).  Each numbered item is an
ASSUMPTION; goldens produced through this stub are "parity unpinned at the PyG
boundary" (the MMD / GradReverse / Attention goldens do not go through it).

 1. ``MessagePassing.propagate(edge_index, **kw)``, aggr='add', flow
    source->target: gather ``x_j = x[edge_index[0]]``, call ``message`` with the
    kwargs its signature names, scatter-add into ``edge_index[1]`` sequentially in
    edge order, then ``update``.
 2. ``add_remaining_self_loops``: drop existing loops, append one loop per node LAST
    in node order, loop weight = existing loop's weight else ``fill_value``.
 3. ``scatter_add(src, index, dim=0, dim_size=N)`` = zeros(N).index_add_(0, index, src).
 4. ``Linear(in, out, bias=False, weight_initializer='glorot')``: weight ``[out,in]``
    ~ U(-a, a), a = sqrt(6/(in+out)), drawn once in ``__init__``;
    ``reset_parameters()`` draws again.  ``inits.glorot/zeros`` likewise.
 5. ``GCNConv`` = lin(no bias, glorot) -> gcn_norm(destination degree, self loops)
    -> one propagate -> + bias(zeros); its ``__init__`` also calls reset_parameters().
 6. ``NeighborLoader(data, [-1]*L, batch_size=N)``: one batch = the whole graph,
    nodes and edges in input order.
 7. ``to_undirected`` = both directions, duplicates merged, sorted by (row, col).
 8. (tdss.py only) ``coalesce`` sorts by (row, col) and drops duplicates; ``spspmm`` returns the
    sparse product sorted by (row, col); ``dense_to_sparse`` lists non-zeros row-major;
    ``remove_self_loops`` masks ``row != col``; ``torch_cluster.random_walk`` steps uniformly
    along row -> col with multiplicity and stays put on a node without out-edges.
 9. (dgsda_base.py only) ``get_laplacian(normalization='sym')`` removes self loops, takes the
    degree over ``row`` and returns ``-D^-1/2 W D^-1/2`` followed by N diagonal ones;
    ``add_self_loops`` appends N loops with the fill value.
11. (graph mode only) ``torch_geometric.loader.DataLoader(dataset, batch_size, shuffle)`` is torch's own
    ``DataLoader`` (real: its RNG draws -- the iterator's base seed, then the RandomSampler's seed, both from the
    default CPU generator -- are the installed torch's) with PyG's collation: ``x`` / ``y`` concatenated in list
    order, ``edge_index`` concatenated with every graph's node ids shifted by the nodes before it, ``batch`` =
    graph index per node, ``num_graphs``.  ``global_mean_pool(x, batch)`` = per-graph sum (index_add in node
    order) divided by the node count (clamped at 1).
12. (graph mode of grade.py / dane.py only) ``len(batch)`` of a collated ``Batch`` = its number of graphs (PyG's
    ``Batch.__len__`` returns ``num_graphs``): grade.py:174,180 size the domain labels / the MMD rows by it.
13. (gnn_base.py:72-95, ``gnn='sage' | 'gin' | 'gat'``)  The three convolutions, per PyG >= 2.5 sources:
    a. PyG ``Linear(in, out, bias, weight_initializer=None)``: weight ~ U(-b, b) with b = 1/sqrt(in)
       (``inits.kaiming_uniform(fan=in, a=sqrt(5))`` = sqrt(6 / ((1 + 5) in))), bias ~ U(-1/sqrt(in), 1/sqrt(in))
       (``inits.uniform``); drawn weight-then-bias in ``Linear.__init__`` (it ends in ``reset_parameters()``) and again
       by every ``reset_parameters()``.
    b. ``SAGEConv(in, out)`` (aggr='mean', root_weight=True, project=False, normalize=False, bias=True):
       ``lin_l = Linear(in, out, bias=True)``, ``lin_r = Linear(in, out, bias=False)``, then ``reset_parameters()``
       (lin_l, lin_r) -- RNG order lin_l.W, lin_l.b, lin_r.W, lin_l.W, lin_l.b, lin_r.W.  forward:
       ``lin_l(mean_{j -> i} x_j) + lin_r(x_i)``; mean = scatter-sum in edge order / max(#messages, 1); multiset
       neighbourhoods, no self loops added, edge weights ignored (the third positional argument GNNBase passes is
       SAGEConv's ``size``: None).
    c. ``GINConv(nn, eps=0., train_eps=True)``: ``eps = Parameter(empty(1))``; ``reset_parameters()`` =
       ``reset(nn)`` (every child's ``reset_parameters()``: the torch ``Linear`` inside the torch ``Sequential``
       gnn_base.py:6 builds draws its weight and bias a SECOND time) then ``eps.fill_(0)``.  forward:
       ``nn(sum_{j -> i} x_j + (1 + eps) x_i)`` (aggregate first, then the root term added).
    d. ``GATConv(in, out, heads=1, concat=False)`` (negative_slope 0.2, dropout 0, add_self_loops=True, bias=True,
       edge_dim=None, residual=False; int ``in``: ONE ``lin = Linear(in, out, bias=False, 'glorot')`` shared by source
       and target, PyG >= 2.5 -- 2.4 names it lin_src / lin_dst and resets it twice): RNG order lin.W (its __init__),
       then ``reset_parameters()``: lin.W, glorot(att_src [1, 1, out]), glorot(att_dst), zeros(bias).  forward:
       h = lin(x); a_s = <h, att_src>, a_d = <h, att_dst>; existing self loops removed and one loop per node appended
       LAST; per edge j -> i: e = leaky_relu(a_s[j] + a_d[i], 0.2); softmax over the incoming edges of i as
       ``exp(e - max_i) / (sum_i exp(e - max_i) + 1e-16)`` (max taken on the detached scores); out_i = sum alpha h_j
       (scatter-sum in edge order), mean over the one head, + bias.
14. (round 6; multi-batch node mode only) ``NeighborLoader(data, [-1]*L, batch_size=B)`` with B < N, ``shuffle=False``:
    seeds are taken B at a time in node order (the last batch is short).  A batch is the L-hop IN-neighbourhood of its
    seeds: hop l expands, in discovery order, every node first reached in hop l-1 (hop 1: the seeds) and takes ALL of
    its in-edges (fan-out -1) in the order of the CSC PyG builds with a stable sort of the edge list by destination
    (``to_csc`` / ``index_sort``: inside one destination, input edge order; duplicate edges are kept as often as they
    occur).  Nodes: the seeds first, then newly reached nodes in discovery order; ``edge_index`` relabelled to that
    numbering, grouped by destination in expansion order (nodes reached in the LAST hop are never expanded: they
    have no in-edges in the batch); ``x`` / ``y`` = the rows of ``n_id``; ``batch_size`` = the number of seeds.
    Deterministic (no draw is made for fan-out -1), which is why the multi-batch goldens use it.
10. (reweight_gnn.py / strurw.py only) ``MessagePassing(aggr='mean', flow='target_to_source')``:
    messages from ``x[edge_index[1]]`` averaged at ``edge_index[0]`` over the number of messages;
    ``update`` receives the propagate kwargs it names; ``to_dense_adj`` sums duplicate edges;
    ``torch_geometric.nn.conv.gcn_conv.gcn_norm`` = assumption 5's normalisation (column degree).
"""
import inspect
import math
import sys
import types
from typing import Optional, Tuple

import torch
import torch.utils.data
from torch import Tensor


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []          # allow sub-imports
    sys.modules[name] = m
    return m


def maybe_num_nodes(edge_index, num_nodes=None):
    if num_nodes is not None:
        return num_nodes
    return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    n = dim_size if dim_size is not None else int(index.max()) + 1
    shape = (n,) + tuple(src.shape[1:])
    return torch.zeros(shape, dtype=src.dtype, device=src.device).index_add_(0, index, src)


def add_remaining_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    N = maybe_num_nodes(edge_index, num_nodes)
    mask = edge_index[0] != edge_index[1]
    loop_index = torch.arange(0, N, dtype=torch.long, device=edge_index.device)
    loop_index = loop_index.unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        loop_attr = edge_attr.new_full((N,) + tuple(edge_attr.shape[1:]), fill_value)
        inv_mask = ~mask
        loop_attr[edge_index[0][inv_mask]] = edge_attr[inv_mask]
        edge_attr = torch.cat([edge_attr[mask], loop_attr], dim=0)
    edge_index = torch.cat([edge_index[:, mask], loop_index], dim=1)
    return edge_index, edge_attr


def to_undirected(edge_index, num_nodes=None):
    """Assumption 7: both directions, duplicates merged, sorted by (row, col) (PyG coalesce)."""
    n = maybe_num_nodes(edge_index, num_nodes)
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    key = torch.unique(both[0] * n + both[1])
    return torch.stack([key // n, key % n])


def is_undirected(edge_index, num_nodes=None):
    n = maybe_num_nodes(edge_index, num_nodes)
    a = torch.unique(edge_index[0] * n + edge_index[1])
    b = torch.unique(edge_index[1] * n + edge_index[0])
    return a.numel() == b.numel() and bool((a == b).all())


def glorot(t):
    if t is not None:
        a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
        t.data.uniform_(-a, a)


def zeros(t):
    if t is not None:
        t.data.fill_(0)


def uniform(size, t):        # imported by mixup_gcnconv.py for a GraphConv class StruRW never builds
    if t is not None:
        bound = 1.0 / math.sqrt(size)
        t.data.uniform_(-bound, bound)


class Linear(torch.nn.Module):
    """Assumptions 4 (glorot, no bias: the GCN-family layers) and 13a (PyG's default initialisers)."""

    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None,
                 bias_initializer=None):
        super().__init__()
        assert weight_initializer in ('glorot', None) and bias_initializer is None
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer = weight_initializer
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight_initializer == 'glorot':
            glorot(self.weight)
        else:                                            # inits.kaiming_uniform(weight, fan=in, a=sqrt(5))
            bound = math.sqrt(6 / ((1 + math.sqrt(5) ** 2) * self.in_channels))
            self.weight.data.uniform_(-bound, bound)
        if self.bias is not None:                        # inits.uniform(in, bias)
            uniform(self.in_channels, self.bias)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight, self.bias)


class MessagePassing(torch.nn.Module):
    """add / mean aggregation; ``flow='target_to_source'`` swaps the roles of the two edge rows
    (x_j = x[edge_index[1]], aggregated at edge_index[0]) -- reweight_gnn.py only.  'mean' divides the
    per-node sum by the number of incoming messages (nodes without messages stay zero)."""

    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2, **kwargs):
        super().__init__()
        assert aggr in ('add', 'mean') and flow in ('source_to_target', 'target_to_source')
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        x = kwargs['x']
        n = x.size(0)
        j, i = (0, 1) if self.flow == 'source_to_target' else (1, 0)
        params = inspect.signature(self.message).parameters
        margs = {}
        for name in params:
            if name == 'x_j':
                margs[name] = x.index_select(0, edge_index[j])
            elif name == 'x_i':
                margs[name] = x.index_select(0, edge_index[i])
            elif name == 'edge_index':
                margs[name] = edge_index
            else:
                margs[name] = kwargs.get(name)
        msg = self.message(**margs)
        out = torch.zeros((n,) + tuple(msg.shape[1:]), dtype=msg.dtype).index_add_(0, edge_index[i], msg)
        if self.aggr == 'mean':
            cnt = torch.zeros(n, dtype=msg.dtype).index_add_(0, edge_index[i], torch.ones(edge_index.size(1), dtype=msg.dtype))
            out = out / cnt.clamp(min=1).view(-1, 1)
        uparams = [k for k in inspect.signature(self.update).parameters if k != 'aggr_out']
        return self.update(out, **{k: kwargs.get(k) for k in uparams})

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def _gcn_norm(edge_index, edge_weight, num_nodes, improved=False, add_self_loops=True):
    fill = 2. if improved else 1.
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1),), dtype=torch.float32)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill, num_nodes)
    row, col = edge_index[0], edge_index[1]
    deg = scatter_add(edge_weight, col, dim=0, dim_size=num_nodes)
    dis = deg.pow_(-0.5)
    dis.masked_fill_(dis == float('inf'), 0)
    return edge_index, dis[row] * edge_weight * dis[col]


class GCNConv(MessagePassing):
    def __init__(self, in_channels, out_channels, improved=False, cached=False,
                 add_self_loops=True, normalize=True, bias=True, **kwargs):
        super().__init__(aggr='add')
        self.improved = improved
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer='glorot')
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        zeros(self.bias)

    def forward(self, x, edge_index, edge_weight=None):
        edge_index, edge_weight = _gcn_norm(edge_index, edge_weight, x.size(0), self.improved)
        x = self.lin(x)
        out = self.propagate(edge_index, x=x, edge_weight=edge_weight)
        return out + self.bias

    def message(self, x_j, edge_weight):
        return edge_weight.view(-1, 1) * x_j


class SAGEConv(MessagePassing):
    """Assumption 13b."""

    def __init__(self, in_channels, out_channels, aggr='mean', normalize=False, root_weight=True, project=False,
                 bias=True, **kwargs):
        assert aggr == 'mean' and not normalize and root_weight and not project
        super().__init__(aggr='mean')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = Linear(in_channels, out_channels, bias=bias)
        self.lin_r = Linear(in_channels, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x, edge_index, size=None):
        assert size is None                              # GNNBase hands its edge_weight=None in this slot
        out = self.propagate(edge_index, x=x)
        out = self.lin_l(out)
        return out + self.lin_r(x)


def _reset(value):
    """PyG ``nn.inits.reset``."""
    if hasattr(value, 'reset_parameters'):
        value.reset_parameters()
    else:
        for child in value.children() if hasattr(value, 'children') else []:
            _reset(child)


class GINConv(MessagePassing):
    """Assumption 13c."""

    def __init__(self, nn, eps=0., train_eps=False, **kwargs):
        super().__init__(aggr='add')
        self.nn = nn
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.empty(1))
        else:
            self.register_buffer('eps', torch.empty(1))
        self.reset_parameters()

    def reset_parameters(self):
        _reset(self.nn)
        self.eps.data.fill_(self.initial_eps)

    def forward(self, x, edge_index, size=None):
        assert size is None
        out = self.propagate(edge_index, x=x)
        out = out + (1 + self.eps) * x
        return self.nn(out)


class GATConv(torch.nn.Module):
    """Assumption 13d (single head)."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0.0,
                 add_self_loops=True, edge_dim=None, fill_value='mean', bias=True, **kwargs):
        super().__init__()
        assert heads == 1 and not concat and dropout == 0.0 and add_self_loops and edge_dim is None and bias
        self.in_channels, self.out_channels, self.negative_slope = in_channels, out_channels, negative_slope
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer='glorot')
        self.att_src = torch.nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = torch.nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        glorot(self.att_src)
        glorot(self.att_dst)
        zeros(self.bias)

    def forward(self, x, edge_index, edge_attr=None, size=None):
        assert edge_attr is None and size is None
        n, H, C = x.size(0), 1, self.out_channels
        h = self.lin(x).view(-1, H, C)
        alpha_src = (h * self.att_src).sum(dim=-1)
        alpha_dst = (h * self.att_dst).sum(dim=-1)
        edge_index, _ = remove_self_loops(edge_index)
        edge_index, _ = add_self_loops(edge_index, num_nodes=n)
        j, i = edge_index[0], edge_index[1]
        alpha = torch.nn.functional.leaky_relu(alpha_src[j] + alpha_dst[i], self.negative_slope)     # [E, H]
        # torch_geometric.utils.softmax(alpha, index=i, num_nodes=n)
        amax = torch.full((n, H), float('-inf')).scatter_reduce(0, i.view(-1, 1).expand(-1, H), alpha.detach(),
                                                                reduce='amax', include_self=True)
        out = (alpha - amax.index_select(0, i)).exp()
        out_sum = torch.zeros(n, H).index_add_(0, i, out) + 1e-16
        alpha = out / out_sum.index_select(0, i)
        msg = alpha.unsqueeze(-1) * h.index_select(0, j)                                              # [E, H, C]
        agg = torch.zeros(n, H, C).index_add_(0, i, msg)
        return agg.mean(dim=1) + self.bias


def global_mean_pool(x, batch, size=None):
    n = int(batch.max()) + 1 if size is None else size
    s = torch.zeros(n, x.size(1), dtype=x.dtype).index_add_(0, batch, x)
    c = torch.zeros(n, dtype=x.dtype).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return s / c.clamp(min=1).view(-1, 1)


class Data:
    edge_attr = None         # PyG's Data answers None for the attributes it declares

    def __init__(self, x=None, edge_index=None, y=None, **kw):
        self.x, self.edge_index, self.y = x, edge_index, y
        self._n = None
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device):
        return self

    @property
    def num_nodes(self):
        return self._n if self._n is not None else self.x.size(0)

    @num_nodes.setter
    def num_nodes(self, n):
        self._n = n


# --- what pygda/models/tdss.py pulls in (assumption 8) -------------------------------
def remove_self_loops(edge_index, edge_attr=None):
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def coalesce(edge_index, edge_attr=None, num_nodes=None, reduce='sum'):
    """PyG ``coalesce``: sort by (row, col), drop duplicates; ``(edge_index, None)`` when the
    attribute argument is given as None (tdss.py:78 passes ``N`` in the ``reduce`` slot: unused)."""
    assert edge_attr is None
    n = maybe_num_nodes(edge_index, num_nodes)
    key = torch.unique(edge_index[0] * n + edge_index[1])          # sorted
    return torch.stack([key // n, key % n]), None


def spspmm(index_a, value_a, index_b, value_b, m, k, n, coalesced=False):
    """torch_sparse ``spspmm``: the sparse product, entries sorted by (row, col)."""
    a = torch.zeros(m, k).index_put_((index_a[0], index_a[1]), value_a, accumulate=True)
    b = torch.zeros(k, n).index_put_((index_b[0], index_b[1]), value_b, accumulate=True)
    c = a @ b
    idx = (c != 0).nonzero().t().contiguous()
    return idx, c[idx[0], idx[1]]


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None):
    """PyG ``to_dense_adj`` for a single graph: ``[1, N, N]``, duplicate edges summed."""
    n = maybe_num_nodes(edge_index, max_num_nodes)
    adj = torch.zeros(n, n).index_put_((edge_index[0], edge_index[1]), torch.ones(edge_index.size(1)), accumulate=True)
    return adj.unsqueeze(0)


def dense_to_sparse(adj):
    """PyG ``dense_to_sparse``: non-zeros in row-major order."""
    idx = adj.nonzero().t().contiguous()
    return idx, adj[idx[0], idx[1]]


def add_self_loops(edge_index, edge_attr=None, fill_value=1., num_nodes=None):
    """PyG ``add_self_loops``: N loops appended last, existing ones untouched."""
    n = maybe_num_nodes(edge_index, num_nodes)
    loops = torch.arange(n, dtype=edge_index.dtype)
    ei = torch.cat([edge_index, torch.stack([loops, loops])], dim=1)
    if edge_attr is None:
        return ei, None
    return ei, torch.cat([edge_attr, torch.full((n,), float(fill_value), dtype=edge_attr.dtype)])


def get_laplacian(edge_index, edge_weight=None, normalization=None, dtype=None, num_nodes=None):
    """PyG ``get_laplacian``: self loops removed, degree over row; 'sym': -D^-1/2 W D^-1/2 plus N
    diagonal ones; None: -W plus the degrees."""
    edge_index, edge_weight = remove_self_loops(edge_index, edge_weight)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype or torch.float32)
    n = maybe_num_nodes(edge_index, num_nodes)
    row, col = edge_index
    deg = torch.zeros(n, dtype=edge_weight.dtype).index_add_(0, row, edge_weight)
    if normalization is None:
        return add_self_loops(edge_index, -edge_weight, 1., n)[0], torch.cat([-edge_weight, deg])
    assert normalization == 'sym'
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float('inf'), 0)
    w = dis[row] * edge_weight * dis[col]
    return add_self_loops(edge_index, -w, 1., n)


def random_walk(row, col, start, walk_length, p=1, q=1, coalesced=True, num_nodes=None):
    """torch_cluster ``random_walk``: uniform step along row -> col over the (row, col)-sorted
    edge list with multiplicity; a node without out-edges stays put.  (Draws from torch's CPU
    generator here; torch_cluster has its own stream, so RW fixtures store the resulting graph.)"""
    n = int(max(row.max(), col.max(), start.max())) + 1 if num_nodes is None else num_nodes
    perm = torch.argsort(row * n + col)
    row, col = row[perm], col[perm]
    deg = torch.zeros(n, dtype=torch.long).index_add_(0, row, torch.ones_like(row))
    ptr = torch.zeros(n + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(deg, 0)
    walk = [start]
    cur = start
    for _ in range(walk_length):
        d = deg[cur]
        pick = (torch.rand(cur.numel()) * d).long().clamp(max=(d - 1).clamp(min=0))
        nxt = torch.where(d > 0, col[(ptr[cur] + pick).clamp(max=max(col.numel() - 1, 0))], cur)
        walk.append(nxt)
        cur = nxt
    return torch.stack(walk, dim=1)


class NeighborLoader:
    """Full batch (assumption 6), or seed mini-batches with whole neighbourhoods (fan-out -1, assumption 14)."""

    def __init__(self, data, num_neighbors, batch_size=1, **kw):
        assert all(k == -1 for k in num_neighbors), "the stub samples nothing: fan-out -1 only"
        assert not kw.get("shuffle", False)
        self.data, self.hops, self.batch_size = data, len(num_neighbors), int(batch_size)

    def __len__(self):
        n = self.data.x.size(0)
        return 1 if self.batch_size >= n else -(-n // self.batch_size)

    def __iter__(self):
        data = self.data
        n = data.x.size(0)
        if self.batch_size >= n:
            yield data
            return
        ei = data.edge_index
        perm = torch.argsort(ei[1], stable=True)                  # CSC: by destination, input order inside one
        src = ei[0][perm].tolist()
        ptr = [0] + torch.cumsum(torch.bincount(ei[1], minlength=n), 0).tolist()
        for start in range(0, n, self.batch_size):
            nodes = list(range(start, min(start + self.batch_size, n)))
            n_seeds = len(nodes)
            local = {v: i for i, v in enumerate(nodes)}
            frontier, rows, cols = list(nodes), [], []
            for _ in range(self.hops):
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
            b = Data(x=data.x[n_id], edge_index=torch.tensor([rows, cols], dtype=torch.long).reshape(2, -1),
                     y=data.y[n_id])
            b.n_id, b.batch_size = n_id, n_seeds
            yield b


class Batch(Data):
    """A collated batch: ``len()`` = number of graphs (assumption 12)."""

    def __len__(self):
        return self.num_graphs


def collate_graphs(graphs):
    """PyG ``Batch.from_data_list`` for the attributes pygda's graph mode reads (assumption 11)."""
    counts = torch.tensor([g.x.size(0) for g in graphs], dtype=torch.long)
    offs = torch.cumsum(counts, 0) - counts
    b = Batch(x=torch.cat([g.x for g in graphs], dim=0),
             edge_index=torch.cat([g.edge_index + o for g, o in zip(graphs, offs.tolist())], dim=1),
             y=torch.cat([g.y.reshape(-1) for g in graphs], dim=0))
    b.batch = torch.repeat_interleave(torch.arange(len(graphs)), counts)
    b.num_graphs = len(graphs)
    return b


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, **kw):
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=collate_graphs, **kw)


def install():
    """Register the stand-ins in ``sys.modules`` (idempotent)."""
    if 'torch_geometric' in sys.modules and getattr(sys.modules['torch_geometric'], '_gda_stub', False):
        return
    tg = _mod('torch_geometric')
    tg._gda_stub = True
    typing_m = _mod('torch_geometric.typing')
    typing_m.Adj = Tensor
    typing_m.OptTensor = Optional[Tensor]
    typing_m.PairTensor = Tuple[Tensor, Tensor]
    typing_m.OptPairTensor = Tuple[Tensor, Optional[Tensor]]
    typing_m.Size = Optional[Tuple[int, int]]
    typing_m.NoneType = type(None)
    nn_m = _mod('torch_geometric.nn')
    nn_m.global_mean_pool = global_mean_pool
    nn_m.GCNConv = GCNConv
    nn_m.MessagePassing = MessagePassing
    nn_m.SAGEConv, nn_m.GATConv, nn_m.GINConv = SAGEConv, GATConv, GINConv
    inits = _mod('torch_geometric.nn.inits')
    inits.glorot, inits.zeros, inits.uniform = glorot, zeros, uniform
    dense = _mod('torch_geometric.nn.dense')
    lin = _mod('torch_geometric.nn.dense.linear')
    lin.Linear = Linear
    dense.Linear = Linear
    conv = _mod('torch_geometric.nn.conv')
    conv.MessagePassing = MessagePassing
    conv.GCNConv = GCNConv
    utils = _mod('torch_geometric.utils')
    utils.add_remaining_self_loops = add_remaining_self_loops
    utils.is_undirected, utils.to_undirected = is_undirected, to_undirected
    utils.remove_self_loops, utils.coalesce, utils.dense_to_sparse = remove_self_loops, coalesce, dense_to_sparse
    utils.add_self_loops, utils.get_laplacian = add_self_loops, get_laplacian
    utils.to_dense_adj = to_dense_adj
    gconv = _mod('torch_geometric.nn.conv.gcn_conv')
    gconv.gcn_norm = lambda ei, ew=None, num_nodes=None, improved=False, add_self_loops=True, flow="source_to_target", dtype=None: \
        _gcn_norm(ei, ew, num_nodes, improved, add_self_loops)
    typing_m.OptTensor = Optional[Tensor]
    tc = _mod('torch_cluster')
    tc.random_walk = random_walk
    nn_utils = _mod('torch_geometric.utils.num_nodes')
    nn_utils.maybe_num_nodes = maybe_num_nodes
    loader = _mod('torch_geometric.loader')
    loader.NeighborLoader, loader.DataLoader = NeighborLoader, DataLoader
    data = _mod('torch_geometric.data')
    data.Data = Data
    ts = _mod('torch_scatter')
    ts.scatter_add = scatter_add
    tsp = _mod('torch_sparse')

    class SparseTensor:      # never instantiated: pygda callers pass edge_index tensors
        pass

    tsp.SparseTensor = SparseTensor
    for name in ('matmul', 'fill_diag', 'sum', 'mul'):
        setattr(tsp, name, None)
    tsp.matmul = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError('SparseTensor inputs are never used'))
    tsp.spspmm = spspmm
