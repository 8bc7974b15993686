"""CPU: the native host neighbour sampler (csrc/gda_sampler.cpp) -- structural parity with
PyG's NeighborLoader contract: seeds first, discovery order, at most k in-neighbours per
frontier node without replacement, exact L-hop in-neighbourhood for fan-out -1, reproducible."""
import numpy as np
import pytest
import torch

from pygda_amd.data import Data, NeighborLoader
from pygda_amd.sampler import NeighborSampler


def rand_graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (2, e), generator=g)


def in_neighbours(ei, n):
    nb = [[] for _ in range(n)]
    for s, d in ei.t().tolist():
        nb[d].append(s)
    return nb


def test_full_fanout_is_exact_khop():
    n, ei = 200, rand_graph(200, 900, 1)
    nb = in_neighbours(ei, n)
    S = NeighborSampler(ei, n)
    seeds = torch.tensor([3, 77, 150])
    n_id, sub = S.sample(seeds, [-1, -1], seed=5)
    assert n_id[:3].tolist() == seeds.tolist() and len(set(n_id.tolist())) == n_id.numel()
    # expected: BFS over in-edges, two hops, every edge into a frontier node kept exactly once
    reach, frontier, want_edges = list(seeds.tolist()), list(seeds.tolist()), []
    for _ in range(2):
        nxt = []
        for v in frontier:
            for u in nb[v]:
                want_edges.append((u, v))
                if u not in reach:
                    reach.append(u); nxt.append(u)
        frontier = nxt
    assert n_id.tolist() == reach                                  # discovery order
    got_edges = [(n_id[a].item(), n_id[b].item()) for a, b in sub.t().tolist()]
    assert got_edges == want_edges                                 # grouped by destination, edge order


@pytest.mark.parametrize("fan", [[5, 3], [15, 10], [1]])
def test_fanout_bounds_and_no_replacement(fan):
    n, ei = 500, rand_graph(500, 6000, 2)
    nb = in_neighbours(ei, n)
    S = NeighborSampler(ei, n)
    seeds = torch.arange(0, 64)
    n_id, sub = S.sample(seeds, fan, seed=9)
    assert n_id[:64].tolist() == list(range(64))
    assert sub.max() < n_id.numel()
    # every sampled edge exists in the graph; per destination at most k distinct picks per hop
    edges = set(map(tuple, ei.t().tolist()))
    per_dst = {}
    for a, b in sub.t().tolist():
        assert (n_id[a].item(), n_id[b].item()) in edges
        per_dst.setdefault(b, []).append(a)
    hop_of = {}
    # destination nodes of hop h are the nodes discovered in hop h-1: seeds are hop-1 destinations
    for b, srcs in per_dst.items():
        deg = len(nb[n_id[b].item()])
        assert len(srcs) <= max(fan) and len(srcs) <= deg
        if deg <= min(fan):
            assert len(srcs) == deg
    # reproducible for a seed, different for another
    n2, s2 = S.sample(seeds, fan, seed=9)
    assert torch.equal(n_id, n2) and torch.equal(sub, s2)
    n3, s3 = S.sample(seeds, fan, seed=10)
    assert not (n3.numel() == n_id.numel() and torch.equal(s3, sub))


@pytest.mark.parametrize("fan,loops", [([-1, -1], False), ([4, 3], False), ([5, 5], True), ([2], True)])
def test_sampler_built_csr_is_gcn_norm_of_the_batch(fan, loops):
    """gda_sampler_csr_norm: the normalised adjacency of a sampled batch built by the sampler -- rows by destination
    and by source, edge order inside a row, the self loop last, w = dis[src] * dis[dst] -- against the oracle's
    gcn_norm (prop_gcn_conv.py:64-81) of the batch's edge list, bit for bit; with and without self-loop edges in
    the input graph (add_remaining_self_loops drops them and re-appends one per node)."""
    from oracle import pygda_cpu as O
    n, ei = 300, rand_graph(300, 2500, 7)
    if loops:
        ei = torch.cat([ei, torch.arange(0, n, 3).repeat(2, 1)], dim=1)[:, torch.randperm(ei.size(1) + n // 3 + (n % 3 > 0),
                                                                                         generator=torch.Generator().manual_seed(1))]
    S = NeighborSampler(ei, n)
    n_id, sub, block = S.sample(torch.arange(10, 50), fan, seed=3, csr=True)
    nb, ne = n_id.numel(), sub.size(1)
    o = S._csr_offsets(nb, ne)
    block = block.numpy()
    cap = nb + ne
    rowptr, t_rowptr = block[o[0]:o[0] + nb + 1], block[o[1]:o[1] + nb + 1]
    colidx, t_colidx = block[o[2]:o[2] + cap], block[o[3]:o[3] + cap]
    val, t_val = block[o[4]:o[4] + cap].view(np.float32), block[o[5]:o[5] + cap].view(np.float32)
    nei, nw = O.gcn_norm(sub, None, nb)                    # kept edges in order, then one loop per node
    nnz = nei.size(1)
    assert rowptr[nb] == nnz and t_rowptr[nb] == nnz and all(o_ % 4 == 0 for o_ in o)
    for key, other, rp, ci, va in ((1, 0, rowptr, colidx, val), (0, 1, t_rowptr, t_colidx, t_val)):
        order = np.argsort(nei[key].numpy(), kind="stable")
        want_rp = np.zeros(nb + 1, dtype=np.int64)
        np.cumsum(np.bincount(nei[key].numpy(), minlength=nb), out=want_rp[1:])
        np.testing.assert_array_equal(rp, want_rp)
        np.testing.assert_array_equal(ci[:nnz], nei[other].numpy()[order])
        np.testing.assert_array_equal(va[:nnz], nw.numpy()[order])


def test_loader_batches_and_rank_sharding():
    n, ei = 300, rand_graph(300, 2000, 3)
    d = Data(x=torch.randn(n, 7), edge_index=ei, y=torch.arange(n) % 5)
    loader = NeighborLoader(d, [4, 4], batch_size=64)
    batches = list(loader)
    assert len(batches) == len(loader) == 5
    seen = []
    for b in batches:
        k = b.batch_size
        assert torch.equal(b.x, d.x[b.n_id]) and torch.equal(b.y, d.y[b.n_id])
        assert b.edge_index.max() < b.x.size(0)
        seen += b.n_id[:k].tolist()
    assert seen == list(range(n))                                   # every seed exactly once, in order
    # two ranks: disjoint seed batches, equal step counts (rank 1 wraps around for the odd batch)
    l0 = NeighborLoader(d, [4, 4], batch_size=64, rank=0, world_size=2)
    l1 = NeighborLoader(d, [4, 4], batch_size=64, rank=1, world_size=2)
    assert len(l0) == len(l1) == 3
    s0 = [b.n_id[:b.batch_size].tolist() for b in l0]
    s1 = [b.n_id[:b.batch_size].tolist() for b in l1]
    assert s0[0][0] == 0 and s1[0][0] == 64 and s0[1][0] == 128 and s1[1][0] == 192 and s0[2][0] == 256
    assert s1[2][0] == 0                                            # (1 + 2*2) % 5: wrapped


def test_bad_seed_rejected():
    S = NeighborSampler(rand_graph(10, 20, 4), 10)
    with pytest.raises(Exception):
        S.sample(torch.tensor([11]), [2])


def test_native_ppmi_builder_matches_oracle_statistically():
    """csrc/gda_ppmi.cpp against the oracle's replay of ppmi_conv.py:98-169.  The generators
    differ, so parity is statistical: with many passes both estimators converge to the same
    PPMI weights; with the reference's 40 passes the summary statistics agree."""
    import numpy as np
    from oracle import pygda_cpu as O
    from pygda_amd.nn.ppmi_conv import ppmi_edges
    g = torch.Generator().manual_seed(5)
    n = 120
    ei = torch.randint(0, n - 1, (2, 260), generator=g)          # node n-1 isolated
    ei = ei[:, ei[0] != ei[1]]

    def as_dict(e, w):
        return {(int(a), int(b)): float(x) for (a, b), x in zip(e.t().tolist(), w.tolist())}

    def agreement(x, y):
        common = [k for k in x if k in y and (x[k] > 0 or y[k] > 0)]
        a, b = np.array([x[k] for k in common]), np.array([y[k] for k in common])
        return np.corrcoef(a, b)[0, 1], len(common), a.mean(), b.mean()

    np.random.seed(0)
    ref = as_dict(*O.ppmi_raw_edges(ei, path_len=5, passes=600))
    np.random.seed(7)
    ref2 = as_dict(*O.ppmi_raw_edges(ei, path_len=5, passes=600))
    mine = as_dict(*ppmi_edges(ei, n, path_len=5, passes=600, seed=1))
    ceiling, _, _, _ = agreement(ref, ref2)            # two runs of the reference algorithm itself
    corr, n_common, ma_, mb_ = agreement(ref, mine)
    assert n_common > 200 and ceiling > 0.95
    assert corr > ceiling - 0.02                        # as close to the reference as it is to itself
    assert abs(ma_ - mb_) < 0.03 * ma_
    # the reference setting: 40 passes
    np.random.seed(1)
    r40 = as_dict(*O.ppmi_raw_edges(ei, path_len=5))
    m40 = as_dict(*ppmi_edges(ei, n, path_len=5, seed=2))
    ra, ma = np.array(list(r40.values())), np.array(list(m40.values()))
    assert abs(len(r40) - len(m40)) < 0.05 * len(r40)                       # same support size
    assert abs((ra > 0).mean() - (ma > 0).mean()) < 0.05
    assert abs(ra[ra > 0].mean() - ma[ma > 0].mean()) < 0.08 * ra[ra > 0].mean()
    assert all(a != n - 1 and b != n - 1 for a, b in m40)                  # isolated node: no walks
    # reproducible, and governed by np.random.seed when no explicit seed is given
    np.random.seed(3); e1, w1 = ppmi_edges(ei, n, 5)
    np.random.seed(3); e2, w2 = ppmi_edges(ei, n, 5)
    assert torch.equal(e1, e2) and torch.equal(w1, w2)


def test_sampled_batches_are_pinned():
    """Digests of three batches on a fixed graph (seeds, fan-outs, RNG seeds fixed): the sampler's draws are keyed on
    (seed, hop, node) and its numbering is discovery order, so a batch is a pure function of its arguments --
    independent of the thread count and of how the relabelling is implemented (N-sized label arrays in round 1, a
    batch-local map since; both produced these digests)."""
    import hashlib
    g = torch.Generator().manual_seed(123)
    n, e = 5000, 60000
    ei = torch.randint(0, n, (2, e), generator=g)
    seeds = torch.randint(0, n, (64,), generator=g)
    want = {((15, 10), 1): (3753, 7042, "0026048238023cad6cc2052ad5c720a8e8f9dec0"),
            ((4, 4, 4), 2): (3007, 4486, "205d423fb5cad5a1dbd575f9ee688653b2170a5c"),
            ((-1,), 3): (762, 770, "b3cee7c84ca0624bcfc695d4731097d8592e3cb3")}
    for threads in (1, 3):
        S = NeighborSampler(ei, n, threads=threads)
        for (fan, seed), (nn, ne, digest) in want.items():
            n_id, sub = S.sample(seeds, list(fan), seed=seed)
            assert (n_id.numel(), sub.size(1)) == (nn, ne)
            assert hashlib.sha1(n_id.numpy().tobytes() + sub.numpy().tobytes()).hexdigest() == digest, (fan, seed, threads)


def test_fanout_minus_one_batches_are_the_stub_loaders_batches():
    """The multi-batch goldens (``a2gnn_fit3_mb_*.npz``) were recorded from the reference's loop over
    ``tests/golden/_pyg_stub.NeighborLoader`` (assumption 14: PyG's published fan-out -1 algorithm -- seeds in node
    order, hop-wise expansion in discovery order, in-edges in stable CSC order).  The product's loader and the oracle's
    ``neighbor_batches`` hand out the SAME batches, bit for bit: node order (the MMD's row draws index it), edge order
    (fp32 summation order), features, labels, seed counts -- on graphs with duplicate edges, self loops and an isolated
    node, with a short last batch."""
    from oracle import pygda_cpu as O
    from pygda_amd.data import Data, NeighborLoader
    from tests.conftest import T, load_golden
    from tests.golden import _pyg_stub as S
    g = load_golden("a2gnn_fit3_mb_mmd")
    for dom in ("src", "tgt"):
        x, ei, y = T(g[f"{dom}_x"]), T(g[f"{dom}_ei"]), T(g[f"{dom}_y"])
        stub = list(S.NeighborLoader(S.Data(x=x, edge_index=ei, y=y), [-1, -1], batch_size=int(g["batch_size"])))
        ours = list(NeighborLoader(Data(x=x, edge_index=ei, y=y), [-1, -1], batch_size=int(g["batch_size"])))
        orac = O.neighbor_batches(O.Graph(x, ei, y), 2, int(g["batch_size"]))
        assert [b.x.size(0) for b in stub] == g[f"{dom}_batch_nodes"].tolist()
        assert len(stub) == len(ours) == len(orac) > 1
        for a, b, c in zip(stub, ours, orac):
            assert torch.equal(a.n_id, b.n_id) and a.batch_size == b.batch_size
            assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.edge_index, c.edge_index)
            assert torch.equal(a.x, b.x) and torch.equal(a.x, c.x) and torch.equal(a.y, b.y) and torch.equal(a.y, c.y)
