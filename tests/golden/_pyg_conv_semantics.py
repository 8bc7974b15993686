"""Second, independent restatement of the three PyG convolutions behind ``GNNBase(gnn='sage'|'gat'|'gin')``
(pygda/nn/gnn_base.py:72-95), written as per-node Python loops over incoming-edge lists -- no tensor
scatter, no dense adjacency -- so that it shares no code path with ``oracle/pygda_cpu.py`` (vectorised
index_add) nor with the dense-algebra check in tests/test_oracle_golden.py.  PyG itself cannot run here
(not vendored, not installable): these are the PUBLISHED semantics of torch_geometric >= 2.4 the build
assumes, numbered like tests/golden/_pyg_stub.py's so a reviewer can check each against PyG's docs.
"Parity unpinned at the PyG boundary" still holds for these three operators.

 S1  SAGEConv(in, out) defaults: aggr='mean', root_weight=True, normalize=False, project=False, bias=True.
     out_i = lin_l(mean_{j in N_in(i)} x_j) + lin_r(x_i); lin_l = Linear(in, out, bias=True),
     lin_r = Linear(in, out, bias=False).  N_in(i) = sources of the edges whose TARGET is i
     (flow 'source_to_target': edge_index[0] = j, edge_index[1] = i), as a MULTISET (duplicate edges count
     twice); no self loops are added; a node without incoming edges aggregates 0.
 S2  GINConv(nn, eps=0., train_eps=True): out_i = nn((1 + eps) * x_i + sum_{j in N_in(i)} x_j), eps a
     learnable scalar initialised to 0; multiset sum; no self loops added.
 S3  GATConv(in, out, heads=1, concat=False, negative_slope=0.2, dropout=0., add_self_loops=True, bias=True,
     edge_dim=None): h = lin(x) with lin = Linear(in, out, bias=False) shared by source and target
     (in_channels is an int); add_self_loops: existing self loops are REMOVED, then exactly one loop per node
     is appended; e_ij = leaky_relu(<h_j, att_src> + <h_i, att_dst>, 0.2) per edge j -> i; alpha = softmax of e
     over the incoming edges of i (multiset: a duplicate edge is its own softmax entry); out_i = sum_j alpha_ij
     h_j, mean over the single head (= identity), + bias.
 S4  All three ignore ``edge_weight`` (GNNBase passes none for sage/gat/gin).
"""
import math

import torch


def _incoming(edge_index, n):
    inc = [[] for _ in range(n)]
    for j, i in zip(edge_index[0].tolist(), edge_index[1].tolist()):
        inc[i].append(j)
    return inc


def sage_loop(x, edge_index, w_l, b_l, w_r):                                   # S1
    n = x.size(0)
    out = torch.zeros(n, w_l.size(0), dtype=torch.float64)
    for i, nb in enumerate(_incoming(edge_index, n)):
        mean = torch.zeros(x.size(1), dtype=torch.float64)
        for j in nb:
            mean += x[j].double()
        if nb:
            mean /= len(nb)
        out[i] = w_l.double() @ mean + b_l.double() + w_r.double() @ x[i].double()
    return out


def gin_loop(x, edge_index, eps, w, b):                                        # S2 with nn = Linear(w, b)
    n = x.size(0)
    out = torch.zeros(n, w.size(0), dtype=torch.float64)
    for i, nb in enumerate(_incoming(edge_index, n)):
        acc = (1.0 + float(eps)) * x[i].double()
        for j in nb:
            acc = acc + x[j].double()
        out[i] = w.double() @ acc + b.double()
    return out


def gat_loop(x, edge_index, w, att_src, att_dst, bias, negative_slope=0.2):    # S3
    n = x.size(0)
    h = [w.double() @ x[i].double() for i in range(n)]
    a_s = [float(h[i] @ att_src.double().view(-1)) for i in range(n)]
    a_d = [float(h[i] @ att_dst.double().view(-1)) for i in range(n)]
    out = torch.zeros(n, w.size(0), dtype=torch.float64)
    for i, nb in enumerate(_incoming(edge_index, n)):
        nb = [j for j in nb if j != i] + [i]                   # loops removed, exactly one appended
        e = []
        for j in nb:
            v = a_s[j] + a_d[i]
            e.append(v if v > 0 else negative_slope * v)
        m = max(e)
        p = [math.exp(v - m) for v in e]
        z = sum(p)
        acc = torch.zeros(w.size(0), dtype=torch.float64)
        for j, pj in zip(nb, p):
            acc += (pj / z) * h[j]
        out[i] = acc + bias.double()
    return out
