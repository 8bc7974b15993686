"""GPU: the device neighbour sampler (csrc/gda_dsampler.hip) against the native host sampler it must reproduce bit
for bit (same counter-based draws keyed on (seed, hop, node), same discovery order), its CSR pair against the device
ingestion of the batch's edge list, and the loader path built on it (missing item 2;
pygda/models/a2gnn.py:260-277 is the call site both samplers stand in for)."""
import hashlib

import numpy as np
import pytest
import torch

from pygda_amd import _lib, ops
from pygda_amd.data import Data, NeighborLoader
from pygda_amd.graph import build_csr
from pygda_amd.sampler import DeviceNeighborSampler, NeighborSampler
from tests.test_gpu_parity import DEV, exact

pytestmark = pytest.mark.gpu


def _graph(n, e, seed, loops=False, multi=False):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    if loops:
        ei = torch.cat([ei, torch.arange(0, n, 3).repeat(2, 1)], dim=1)
    if multi:
        ei = torch.cat([ei, ei[:, : e // 10]], dim=1)
    return ei[:, torch.randperm(ei.size(1), generator=g)]


def test_device_batches_match_the_pinned_digests():
    """The digests of tests/test_sampler_host.py::test_sampled_batches_are_pinned, produced on the device."""
    g = torch.Generator().manual_seed(123)
    n, e = 5000, 60000
    ei = torch.randint(0, n, (2, e), generator=g)
    seeds = torch.randint(0, n, (64,), generator=g)
    want = {((15, 10), 1): (3753, 7042, "0026048238023cad6cc2052ad5c720a8e8f9dec0"),
            ((4, 4, 4), 2): (3007, 4486, "205d423fb5cad5a1dbd575f9ee688653b2170a5c"),
            ((-1,), 3): (762, 770, "b3cee7c84ca0624bcfc695d4731097d8592e3cb3")}
    S = DeviceNeighborSampler(ei.to(DEV), n)
    assert S.max_in_degree == int(torch.bincount(ei[1], minlength=n).max())
    for (fan, seed), (nn, ne, digest) in want.items():
        assert S.supports(seeds.numel(), list(fan))
        n_id, sub = S.sample(seeds, list(fan), seed=seed)
        assert (n_id.numel(), sub.size(1)) == (nn, ne)
        got = hashlib.sha1(n_id.cpu().numpy().tobytes() + sub.cpu().contiguous().numpy().tobytes()).hexdigest()
        assert got == digest, (fan, seed)


@pytest.mark.parametrize("fan", [[15, 10], [3, 2, 2], [-1, -1], [2, -1], [-1, 3], [1], [64, 1]])
@pytest.mark.parametrize("variant", ["plain", "loops+multi"])
def test_device_sampler_equals_host_sampler(fan, variant):
    """Same nodes in the same order, same edges in the same order -- on graphs with self-loop edges, repeated edges,
    isolated nodes and hubs, seeds with duplicates -- and the CSR pair equals what the device ingestion builds from
    the batch's edge list, bit for bit (structure and weights, both orientations)."""
    n = 3000
    ei = _graph(n, 30000, 5, loops=variant != "plain", multi=variant != "plain")
    ei = torch.cat([ei, torch.stack([torch.randint(0, n, (400,), generator=torch.Generator().manual_seed(2)),
                                     torch.full((400,), 17)])], dim=1)          # a hub: 400 extra in-neighbours of node 17
    ei = ei[:, ei[1] % 11 != 5]                                                  # nodes without in-neighbours
    H = NeighborSampler(ei, n, threads=2)
    D = DeviceNeighborSampler(ei.to(DEV), n)
    g = torch.Generator().manual_seed(9)
    for trial in range(3):
        seeds = torch.randint(0, n, (97,), generator=g)
        if trial == 1:
            seeds = torch.cat([seeds, seeds[:13], torch.tensor([17, 17])])       # duplicates among the seeds
        hn, he = H.sample(seeds, fan, seed=100 + trial)
        p = D.enqueue(seeds, fan, seed=100 + trial)
        nn, ne, nnz, n_int = p.wait()
        assert 0 < n_int <= nn
        exact(p.nodes[:nn], hn)
        exact(p.ei[:, :ne], he)
        G = build_csr(he.to(DEV), nn)
        assert nnz == G.nnz
        rp, ci, va, trp, tci, tva = p.csr
        exact(rp[:nn + 1], G.rowptr[:nn + 1]); exact(trp[:nn + 1], G.t_rowptr[:nn + 1])
        exact(ci[:nnz], G.colidx[:nnz]); exact(tci[:nnz], G.t_colidx[:nnz])
        exact(va[:nnz].view(torch.int32), G.val[:nnz].view(torch.int32))
        exact(tva[:nnz].view(torch.int32), G.t_val[:nnz].view(torch.int32))
    # reproducible for a seed, different for another (when the fan-out leaves a choice)
    a = D.sample(seeds, fan, seed=7); b = D.sample(seeds, fan, seed=7); c = D.sample(seeds, fan, seed=8)
    exact(a[0], b[0]); exact(a[1], b[1])
    if any(0 < k < 20 for k in fan):
        assert a[1].shape != c[1].shape or not torch.equal(a[1], c[1])


def test_device_sampler_rejects_what_it_cannot_do():
    ei = _graph(50, 400, 1)
    D = DeviceNeighborSampler(ei.to(DEV), 50)
    assert not D.supports(8, [0]) and not D.supports(8, [65]) and D.supports(8, [64, -1])
    with pytest.raises(_lib.GdaError):
        D.sample(torch.tensor([3, 50]), [2])                                     # a seed outside the graph
    with pytest.raises(IndexError):
        DeviceNeighborSampler(torch.tensor([[0, 51], [1, 2]], device=DEV), 50)
    # no edges at all: every batch is its seeds
    Z = DeviceNeighborSampler(torch.empty(2, 0, dtype=torch.int64, device=DEV), 10)
    n_id, sub = Z.sample(torch.tensor([4, 2, 4]), [3, 3])
    exact(n_id, [4, 2]); assert sub.size(1) == 0


@pytest.mark.parametrize("prefetch", [0, 2])
def test_loader_on_the_device_sampler_equals_the_host_loader(monkeypatch, prefetch):
    """NeighborLoader over GPU-resident data: batches (x, y, n_id, edge_index, batch_size) equal those of the host
    sampler path, in order, with and without the prefetching producer thread; the batch carries the prebuilt
    normalised graph, which equals the ingestion of its edge list."""
    from pygda_amd.graph import as_graph
    n = 4000
    ei = _graph(n, 50000, 3, loops=True)
    g = torch.Generator().manual_seed(0)
    d = Data(x=torch.randn(n, 24, generator=g), edge_index=ei, y=torch.randint(0, 5, (n,), generator=g)).to(DEV)
    kw = dict(batch_size=256, input_nodes=torch.randperm(n, generator=g)[:1500], device=DEV, prefetch=prefetch)
    dev_loader = NeighborLoader(d, [6, 4], **kw)
    dev_batches = list(dev_loader)
    assert "device sampler" in dev_loader.sampler_description()
    monkeypatch.setenv("PYGDA_AMD_DEVICE_SAMPLER", "0")
    host_loader = NeighborLoader(d, [6, 4], **kw)
    host_batches = list(host_loader)
    assert "host sampler" in host_loader.sampler_description()
    assert len(dev_batches) == len(host_batches) == 6
    for a, b in zip(dev_batches, host_batches):
        assert a.batch_size == b.batch_size
        exact(a.n_id, b.n_id); exact(a.edge_index, b.edge_index); exact(a.x, b.x); exact(a.y, b.y)
        ga = as_graph(a.edge_index, a.x.size(0))
        assert ga is a.edge_index._gda_prebuilt and ga.transient
        gb = build_csr(b.edge_index.contiguous(), b.x.size(0))
        assert ga.nnz == gb.nnz
        exact(ga.rowptr, gb.rowptr); exact(ga.colidx[:ga.nnz], gb.colidx[:gb.nnz])
        exact(ga.val[:ga.nnz].view(torch.int32), gb.val[:gb.nnz].view(torch.int32))
        exact(ga.t_colidx[:ga.nnz], gb.t_colidx[:gb.nnz])
    # a second epoch draws different neighbourhoods (the epoch enters the RNG seed), same seeds
    again = list(dev_loader)
    exact(again[0].n_id[:256], dev_batches[0].n_id[:256])
    assert again[0].n_id.numel() != dev_batches[0].n_id.numel() or not torch.equal(again[0].n_id, dev_batches[0].n_id)


@pytest.mark.parametrize("lag", [False, True])
def test_recycling_loader_hands_out_the_batches_of_the_allocating_loader(lag):
    """``NeighborLoader(recycle=True)`` (round 5: a ring of prefetch + 4 blocks, one foreign call per batch in the producer
    thread -- sampler._Ring / gda_dsampler_batch) against the loader that allocates every batch, consumed the way the
    trainers consume them (done with a batch when the next one is taken): ids, edge list, features, labels, the
    prebuilt CSR pair, the interior K-step plans and their verdicts, over three times the ring's depth plus a ragged
    last batch (which takes the allocating path) and a second pass.  ``lag``: the consumer's stream is kept a few
    milliseconds behind its host thread, so that blocks come round while the device still reads them -- what the
    ring's ``free`` events order."""
    n = 60000
    ei = _graph(n, 600000, 5, loops=True)
    g = torch.Generator().manual_seed(1)
    d = Data(x=torch.randn(n, 16, generator=g), edge_index=ei, y=torch.randint(0, 5, (n,), generator=g)).to(DEV)
    kw = dict(batch_size=512, input_nodes=torch.randint(0, n, (512 * 20 + 100,), generator=g), device=DEV, prefetch=2)
    fresh, ring = NeighborLoader(d, [5, 4], **kw), NeighborLoader(d, [5, 4], recycle=True, **kw)

    def digest(b):
        G = b.edge_index._gda_prebuilt
        parts = [b.n_id.sum(), b.edge_index.sum(), b.x.double().sum(), b.y.sum(), G.rowptr.sum(), G.colidx[:G.nnz].sum(),
                 G.val[:G.nnz].double().sum(), G.t_rowptr.sum(), G.t_colidx[:G.nnz].sum(), G.t_val[:G.nnz].double().sum()]
        if G.iplan is not None:               # the plans at work (their unused bytes are not defined): forward and transposed
            parts.append(ops.spmm_kstep(G, b.x, 3, None).double().sum())
            parts.append(ops.spmm_kstep(G, b.x, 3, None, transposed=True).double().sum())
        return torch.stack([p.double() for p in parts]), (b.x.size(0), b.edge_index.size(1), G.nnz, G.n_interior,
                                                           tuple(pl is not None for pl in (G.iplan or ())))

    for _ in range(2):                       # second pass: the ring is reset and written again
        want = [digest(b) for b in fresh]
        got = []
        for b in ring:
            if lag:
                torch.cuda._sleep(3_000_000)            # ~1.5 ms of device time ahead of every read of the batch
            got.append(digest(b))
        torch.cuda.synchronize()
        assert len(want) == len(got) == 21
        assert ring._ring is not None and ring._ring.depth == 6 and ring._ring.at == 20        # the ragged batch allocates
        for (a, sa), (b, sb) in zip(want, got):
            assert sa == sb
            exact(a, b)


@pytest.mark.parametrize("K,d", [(1, 128), (3, 5), (10, 128), (4, 36)])
def test_interior_rows_kstep_on_sampled_batches(monkeypatch, K, d):
    """A sampled batch's last-hop discoveries are never expanded: their rows hold a unit self loop only.  K steps that
    recompute the interior rows only (gda_spmm_csr_interior_kstep_f32) equal K full steps -- forward bit for bit
    (signed zeros aside) for K = 1 and whenever the leaf columns are gathered per step, to fp32 summation order when
    their contribution is formed once (K >= 2, the default); transposed (backward) to fp32 summation order -- with and
    without the bias, and through autograd."""
    from pygda_amd import ops
    from pygda_amd.graph import as_graph
    n = 20000
    ei = _graph(n, 300000, 11, loops=True)
    g = torch.Generator().manual_seed(K * 7 + d)
    data = Data(x=torch.randn(n, 16, generator=g), edge_index=ei, y=torch.zeros(n, dtype=torch.long)).to(DEV)
    loader = NeighborLoader(data, [7, 5], batch_size=300, input_nodes=torch.randperm(n, generator=g)[:600], device=DEV)
    for batch in loader:
        G = as_graph(batch.edge_index, batch.x.size(0))
        nb = batch.x.size(0)
        assert G.n_interior is not None and 300 <= G.n_interior < nb // 2
        # rows from n_interior on: exactly their self loop, weight 1
        rp = G.rowptr[:nb + 1].cpu()
        assert bool((rp[G.n_interior + 1:] - rp[G.n_interior:-1] == 1).all())
        x = torch.randn(nb, d, generator=g).to(DEV)
        bias = torch.randn(d, generator=g).to(DEV)
        gy = torch.randn(nb, d, generator=g).to(DEV)
        got = ops.spmm_kstep(G, x, K, bias)
        got_t = ops.spmm_kstep(G, gy, K, None, transposed=True)
        xa = x.clone().requires_grad_()
        ops.propagate(xa, G, K, bias).backward(gy)
        monkeypatch.setattr(ops, "INTERIOR_KSTEP", False)
        want = ops.spmm_kstep(G, x, K, bias)
        want_t = ops.spmm_kstep(G, gy, K, None, transposed=True)
        monkeypatch.setattr(ops, "INTERIOR_KSTEP", True)
        if K >= 2:       # the leaf columns' contribution is formed once and added per step: re-associated row sums
            np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6 * float(want.abs().max()))
            monkeypatch.setattr(ops, "INTERIOR_HOIST", False)
            exact(ops.spmm_kstep(G, x, K, bias), want)          # without the hoist: the full steps' sums bit for bit
            monkeypatch.setattr(ops, "INTERIOR_HOIST", True)
        else:
            exact(got, want)
        scale = float(want_t.abs().max())
        np.testing.assert_allclose(got_t.cpu().numpy(), want_t.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)
        np.testing.assert_allclose(xa.grad.cpu().numpy(), want_t.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)


# ---- the one-launch interior K-step (csrc/gda_interior.inc) ---------------------------------------------------------
_IL_E, _IL_TB, _IL_RS = 26, 1024, 16
_IL_ZERO, _IL_DUMP0, _IL_PW = 0, 64, 128
_IL_LW = _IL_PW + _IL_TB
_IL_HEAD = _IL_LW + _IL_TB


def _il_plan_arrays(plan):
    """The device plan (gda_interior_plan_build) as thread-major numpy arrays: the layout include/gda_hip.h leaves
    opaque, restated here so that the kernel can be checked against an emulation of its own program."""
    b = plan.cpu().numpy()
    off_pw = _IL_TB * _IL_E * 4
    off_diag = 2 * off_pw
    off_act = off_diag + _IL_RS * _IL_TB * 4
    off_lead = off_act + _IL_TB * 4
    off_wmax = off_lead + _IL_TB * 4
    tm = lambda a: np.ascontiguousarray(a.reshape(16, _IL_E, 64).transpose(0, 2, 1).reshape(_IL_TB, _IL_E))
    return dict(pk=tm(b[:off_pw].view(np.uint32)), pw=tm(b[off_pw:off_diag].view(np.float32)),
                diag=b[off_diag:off_act].view(np.float32).copy(), actm=b[off_act:off_lead].view(np.uint32).copy(),
                lead=b[off_lead:off_wmax].view(np.uint32).copy(), wmax=b[off_wmax:off_wmax + 64].view(np.uint32).copy())


def _il_emulate(P, x, c, K, n_int, trans):
    """k_il_lds for ONE feature column in numpy float32 (every multiply and add rounded separately, as the kernel's
    __fmul_rn / __fadd_rn): -> (values of the interior rows after K steps, the owner registers `cc`)."""
    f32 = np.float32
    RS = -(-n_int // _IL_TB)
    bufw = _IL_HEAD + RS * _IL_TB
    cur, nxt = np.zeros(bufw, f32), np.full(bufw, np.nan, f32)
    nxt[_IL_ZERO] = 0
    cur[_IL_HEAD:_IL_HEAD + n_int] = x
    rows = np.arange(RS * _IL_TB)
    dg = P["diag"][:RS * _IL_TB].astype(f32)
    cc = np.zeros(RS * _IL_TB, f32)
    if not trans:
        cc[:n_int] = c
    act = ((P["actm"][rows & (_IL_TB - 1)] >> (rows >> 10)) & 1).astype(bool)
    wm = P["wmax"][np.arange(_IL_TB) >> 6]
    idx, st = (P["pk"] & 0xffff).astype(np.int64), (P["pk"] >> 16).astype(np.int64)
    for _ in range(K):
        acc = np.zeros(_IL_TB, f32)
        for e in range(_IL_E):
            on = e < wm
            acc = np.where(on, acc + P["pw"][:, e] * cur[idx[:, e]], acc).astype(f32)
            nxt[st[on, e]] = acc[on]
            acc = np.where(on & (st[:, e] >= _IL_PW), f32(0), acc).astype(f32)
        for t in np.nonzero(P["lead"])[0]:
            v = f32(0)
            for k in range(int(P["lead"][t] >> 16), 0, -1):
                v = f32(v + nxt[_IL_PW + t - k])
            nxt[int(P["lead"][t] & 0xffff)] = f32(v + nxt[_IL_LW + t])
        a = cur[_IL_HEAD + rows]
        o = np.where(act, nxt[_IL_HEAD + rows], f32(0)).astype(f32)
        v = (o + (dg * a).astype(f32)).astype(f32)
        if trans:
            cc = (cc + a).astype(f32)
        else:
            v = (v + cc).astype(f32)
        nxt[_IL_HEAD + rows] = v
        cur, nxt = nxt, cur
    return cur[_IL_HEAD:_IL_HEAD + n_int].copy(), cc[:n_int].copy()


def _leaf_contribution(G, x, n_int, col):
    """c = A_IL x_L of one feature column in k_il_prep's order: the row's entries in CSR order, leaf columns only."""
    f32 = np.float32
    rp = G.rowptr[:n_int + 1].cpu().numpy().astype(np.int64)
    ci, va = G.colidx.cpu().numpy().astype(np.int64), G.val.cpu().numpy()
    xc = x[:, col].cpu().numpy()
    acc = np.zeros(n_int, f32)
    for k in range(int((rp[1:] - rp[:-1]).max())):
        at = rp[:-1] + k
        ok = at < rp[1:]
        at = np.where(ok, at, 0)
        take = ok & (ci[at] >= n_int)
        acc = np.where(take, acc + (va[at] * xc[ci[at]]).astype(f32), acc).astype(f32)
    return acc


def _plan_problem(name):
    """(graph, fan-outs, seeds per batch) of the regimes the plan builder must handle."""
    if name == "uniform":                         # seeds gather ~fan-out interior rows, nobody else does
        return _graph(200_000, 4_000_000, 21), [15, 10], 1024
    if name == "symmetric":                       # cfg-S in small: half of the last-but-one-hop rows sample their seed back
        ei = _graph(2_000_000, 20_000_000, 24)    # (large enough that other interior neighbours stay rare: T ~ 24 k <= 26,624)
        return torch.cat([ei, ei.flip(0)], dim=1), [15, 10], 1024
    if name == "dense":                           # a small graph: last-but-one-hop rows find interior neighbours too
        return _graph(6000, 90_000, 22, loops=True, multi=True), [7, 5], 300
    g = torch.Generator().manual_seed(23)         # power law: a hub's transposed row crosses many runs
    n, e = 50_000, 1_000_000
    w = torch.arange(1, n + 1, dtype=torch.float64).pow(-0.9)
    ei = torch.stack([torch.multinomial(w, e, True, generator=g), torch.randint(0, n, (e,), generator=g)])
    return ei, [10, 5], 512


@pytest.mark.parametrize("name", ["uniform", "symmetric", "dense", "powerlaw"])
def test_interior_lds_plan_and_step_loop_against_their_emulation(monkeypatch, name):
    """gda_interior_plan_build + gda_interior_kstep_lds_f32 (the K interior steps of a sampled batch in ONE launch of
    the step loop): the plans the sampler's stream built on the device are read back and EMULATED in numpy float32 --
    runs, pieces of rows cut by run boundaries, owner pass -- and the kernel must reproduce the emulation bit for bit
    on every interior row, forward (with the leaf columns' contribution c = A_IL x_L in k_il_prep's order) and
    transposed (with the running sum of the step inputs); leaf rows forward are x + bias exactly.  Then against the
    K-launch chain it replaces and against K full aggregations (fp32 summation order)."""
    from pygda_amd import ops, sampler
    from pygda_amd.graph import as_graph
    assert sampler.INTERIOR_LDS
    ei, fan, nseeds = _plan_problem(name)
    n = int(ei.max()) + 1
    g = torch.Generator().manual_seed(5)
    data = Data(x=torch.randn(n, 8, generator=g), edge_index=ei, y=torch.zeros(n, dtype=torch.long)).to(DEV)
    loader = NeighborLoader(data, fan, batch_size=nseeds, input_nodes=torch.randperm(n, generator=g)[:2 * nseeds], device=DEV)
    L = _lib.lib()
    seen = 0
    for batch in loader:
        G = as_graph(batch.edge_index, batch.x.size(0))
        nb, n_int = batch.x.size(0), G.n_interior
        assert G.iplan is not None and 0 < n_int <= L.gda_interior_max_rows()
        for K, d in ((10, 128), (3, 64)):
            x = torch.randn(nb, d, generator=g).to(DEV)
            bias = torch.randn(d, generator=g).to(DEV)
            gy = torch.randn(nb, d, generator=g).to(DEV)
            got = ops.spmm_kstep(G, x, K, bias)
            got_t = ops.spmm_kstep(G, gy, K, None, transposed=True)
            for trans, plan, inp, out in ((False, G.iplan[0], x, got), (True, G.iplan[1], gy, got_t)):
                assert plan is not None, (name, trans)               # these regimes fit the plan
                P = _il_plan_arrays(plan)
                rp = (G.t_rowptr if trans else G.rowptr)[:n_int + 1].cpu().numpy().astype(np.int64)
                ci = (G.t_colidx if trans else G.colidx).cpu().numpy()
                # the plan holds every off-diagonal interior entry exactly once, in row order
                ent = np.concatenate([ci[a:b] for a, b in zip(rp[:-1], rp[1:])])
                rows_of = np.repeat(np.arange(n_int), rp[1:] - rp[:-1])
                keep = (ent < n_int) & (ent != rows_of)
                T = int(keep.sum())
                q = max(1, -(-T // _IL_TB))
                flat = P["pk"][:, :q].reshape(-1)[:T]
                exact(flat & 0xffff, ent[keep] + _IL_HEAD)
                dump = ((_IL_DUMP0 + (np.arange(_IL_TB) & 63)) << 16).astype(np.uint32)[:, None]
                assert int((P["pk"][:, q:] != (_IL_ZERO | dump)).sum()) == 0        # unused entries: zero word -> the lane's dump word
                for col in (0, d - 1):
                    c = None if trans else _leaf_contribution(G, x, n_int, col)
                    want, cc = _il_emulate(P, inp[:n_int, col].cpu().numpy(), c, K, n_int, trans)
                    if not trans:
                        want = (want + bias[col].cpu().numpy()).astype(np.float32)
                    exact(out[:n_int, col], want)
            exact(got[n_int:], x[n_int:] + bias)                       # leaves: their unit self loop
            # the chain it replaces, and K full aggregations
            monkeypatch.setattr(ops, "_interior_lds_plan", lambda *a, **k: None)
            chain, chain_t = ops.spmm_kstep(G, x, K, bias), ops.spmm_kstep(G, gy, K, None, transposed=True)
            monkeypatch.undo()
            monkeypatch.setattr(ops, "INTERIOR_KSTEP", False)
            full, full_t = ops.spmm_kstep(G, x, K, bias), ops.spmm_kstep(G, gy, K, None, transposed=True)
            monkeypatch.undo()
            for a, b in ((got, chain), (got, full), (got_t, chain_t), (got_t, full_t)):
                np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=2e-6 * float(b.abs().max()))
            # through autograd: backward = the transposed call
            xa = x.clone().requires_grad_()
            ops.propagate(xa, G, K, bias).backward(gy)
            exact(xa.grad, got_t)
        seen += 1
    assert seen == 2


def test_interior_lds_plan_declines_what_does_not_fit_and_the_chain_takes_over():
    """More off-diagonal interior entries than 1024 runs of 26 hold (a dense little graph at fan-out [30, 30]): the
    builder's verdict is 0 for that direction, the graph carries no plan for it, and the K-launch chain runs -- same
    values as K full aggregations."""
    from pygda_amd import ops
    from pygda_amd.graph import as_graph
    n = 1500
    ei = _graph(n, 120_000, 31)
    g = torch.Generator().manual_seed(6)
    data = Data(x=torch.randn(n, 8, generator=g), edge_index=ei, y=torch.zeros(n, dtype=torch.long)).to(DEV)
    loader = NeighborLoader(data, [30, 30], batch_size=700, input_nodes=torch.arange(700), device=DEV)
    batch = next(iter(loader))
    G = as_graph(batch.edge_index, batch.x.size(0))
    assert G.n_interior is not None and G.iplan == (None, None)
    x = torch.randn(batch.x.size(0), 64, generator=g).to(DEV)
    got = ops.spmm_kstep(G, x, 5, None)
    ops.INTERIOR_KSTEP = False
    try:
        want = ops.spmm_kstep(G, x, 5, None)
    finally:
        ops.INTERIOR_KSTEP = True
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6 * float(want.abs().max()))


def test_sampled_training_no_longer_depends_on_host_sampler_threads():
    """cfg-S style training through the trainer: the loaders pick the device sampler, the step trains."""
    import pygda_amd
    from bench import make_cfg_s
    N = 200_000
    src, tgt = make_cfg_s(N, 20, 64, 5, 200, DEV), make_cfg_s(N, 20, 64, 5, 201, DEV)
    m = pygda_amd.models.A2GNN(64, 32, 5, num_layers=2, dropout=0.5, s_pnums=0, t_pnums=10, weight=10, lr=0.005,
                               device=DEV, epoch=1, verbose=0, batch_size=512, num_neigh=[15, 10])
    net, optimizer, step, alpha = m._prepare(src, tgt)
    m.source_loader.input_nodes = m.source_loader.input_nodes[:4 * 512]
    m.target_loader.input_nodes = m.target_loader.input_nodes[:4 * 512]
    seen = []
    m.epoch_hook = lambda e, loss, acc, secs: seen.append((loss, acc))
    torch.manual_seed(1)
    m._train_epochs(net, optimizer, step, alpha)
    assert "device sampler" in m.source_loader.sampler_description()
    assert len(seen) == 1 and np.isfinite(seen[0][0])
    logits, labels = m.predict(tgt)
    assert logits.shape == (4 * 512, 5) and bool(torch.isfinite(logits).all())
    exact(labels, tgt.y[:4 * 512])


def _sampled_fit(monkeypatch, env, steps=6, seed=1, dropout=0.5, extra_seeds=0):
    """cfg-S in small through the trainer's own loop: 200 k nodes per domain, 512 seeds at fan-out [15, 10], dropout on."""
    import pygda_amd
    from bench import make_cfg_s
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N = 200_000
    src, tgt = make_cfg_s(N, 20, 64, 5, 200, DEV), make_cfg_s(N, 20, 64, 5, 201, DEV)
    m = pygda_amd.models.A2GNN(64, 32, 5, num_layers=2, dropout=dropout, s_pnums=0, t_pnums=10, weight=10, lr=0.005,
                               weight_decay=0.001, device=DEV, epoch=2, verbose=0, batch_size=512, num_neigh=[15, 10])
    torch.manual_seed(seed)
    ops.dropout_state.counter(torch.device(DEV)).zero_()
    ops.dropout_state.seed = 12345
    net, optimizer, step, alpha = m._prepare(src, tgt)
    m.source_loader.input_nodes = m.source_loader.input_nodes[:steps * 512 + extra_seeds]
    m.target_loader.input_nodes = m.target_loader.input_nodes[:steps * 512 + extra_seeds]
    seen = []
    m.epoch_hook = lambda e, loss, acc, secs: seen.append((loss, acc))
    m._train_epochs(net, optimizer, step, alpha)
    torch.cuda.synchronize()
    return m, seen, {k: v.detach().clone() for k, v in net.state_dict().items()}


def test_captured_sampled_step_equals_the_eager_static_step_bit_for_bit(monkeypatch):
    """the sampled step captured once at its static capacity shape and replayed
    (pygda_amd/sampled_graph.py) against the SAME static-shape step issued eagerly every time -- same kernels, same
    shapes, same draws, dropout on: per-epoch loss and accuracy and every parameter after 2 x 6 steps, bit for bit.  Then
    against the ordinary eager loop on the batches' real shapes, dropout off (the keep-bits of the stacked source rows are
    keyed on the element index, which moves with the row count; everything else differs by the row counts in the
    reductions: fp32 summation order) -- losses to 1e-5 relative, parameters to 2e-2 in norm (Adam amplifies rounding-level
    gradient differences of near-zero entries)."""
    mc, seen_c, par_c = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "1", "PYGDA_AMD_SAMPLED_GRAPH_CAPTURE": "1"})
    st = mc._sampled_graphed[1]
    assert st.graph is not None and st.replays == 12 and st.fallbacks == 0, (st.replays, st.fallbacks)
    assert mc.source_loader.static_interior == 512 * 16 and st.static[0].n_int == 512 * 16
    assert st.static[0].ncap < st.static[0].ncap_block          # the static shape hugs the live count, not the capacity
    me, seen_e, par_e = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "1", "PYGDA_AMD_SAMPLED_GRAPH_CAPTURE": "0"})
    assert me._sampled_graphed[1].graph is None and me._sampled_graphed[1].replays == 12
    assert seen_c == seen_e, (seen_c, seen_e)
    for k in par_c:
        exact(par_c[k], par_e[k])
    mc, seen_c, par_c = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "1", "PYGDA_AMD_SAMPLED_GRAPH_CAPTURE": "1"},
                                     dropout=0.0)
    assert mc._sampled_graphed[1].replays == 12 and mc._sampled_graphed[1].fallbacks == 0
    mr, seen_r, par_r = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "0"}, dropout=0.0)
    assert getattr(mr, "_sampled_graphed", None) is None
    np.testing.assert_allclose([v[0] for v in seen_c], [v[0] for v in seen_r], rtol=1e-5)
    np.testing.assert_allclose([v[1] for v in seen_c], [v[1] for v in seen_r], atol=1e-4)      # a few rows in 80 k may flip
    for k in par_c:
        # Two runs whose reductions differ in summation order walk apart under Adam (an entry whose gradient is zero up to
        # rounding takes a step of +-lr whatever its size; the split-K products themselves agree with float64 to 4e-7 at both
        # row counts): the losses above are the comparison, the parameters are checked in norm.
        ref = par_r[k].double()
        rel = float((par_c[k].double() - ref).norm() / ref.norm().clamp_min(1e-30))
        assert rel <= 2e-2, (k, rel)
    # predict() after a captured fit: the loaders still hand out ordinary batches
    logits, labels = mc.predict(None)
    assert logits.shape == (6 * 512, 5) and bool(torch.isfinite(logits).all())


def test_captured_sampled_step_with_a_short_last_batch(monkeypatch):
    """Seeds that do not fill the last batch (5 x 512 + 100): that batch has another shape than the ring's blocks, comes as
    an ordinary pending batch and takes the eager step -- per epoch five replays and one fall-back; the epoch's numbers are
    those of the all-eager loop to fp32 summation order, and the ring blocks of replayed and eager steps are all handed
    back (two epochs run through the same six-block ring without touching a batch still in use)."""
    mc, seen_c, par_c = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "1"}, steps=5, dropout=0.0, extra_seeds=100)
    st = mc._sampled_graphed[1]
    assert st.replays == 10 and st.fallbacks == 2, (st.replays, st.fallbacks)
    mr, seen_r, par_r = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "0"}, steps=5, dropout=0.0, extra_seeds=100)
    np.testing.assert_allclose([v[0] for v in seen_c], [v[0] for v in seen_r], rtol=1e-5)
    np.testing.assert_allclose([v[1] for v in seen_c], [v[1] for v in seen_r], atol=1e-4)      # a few rows in 80 k may flip
    for k in par_c:                                       # (in norm: see the test above)
        ref = par_r[k].double()
        rel = float((par_c[k].double() - ref).norm() / ref.norm().clamp_min(1e-30))
        assert rel <= 2e-2, (k, rel)


def test_sampled_training_with_and_without_grad_sinks_is_the_same_run(monkeypatch):
    """ops.GradSink in the trainer (A2GNN's sampled step opens `grad_sinks()`): four activation backward launches per step
    ride in their producers' epilogues.  Same 2 x 6 captured steps, dropout on, with the protocol switched off: per-epoch
    numbers and every parameter bit for bit."""
    hits = ops.sink_hits
    m1, seen1, par1 = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "1"})
    assert ops.sink_hits - hits >= 4, ops.sink_hits - hits        # (counted while capturing / warming up: replays run no Python)
    monkeypatch.setattr(ops, "GRAD_SINKS", False)
    hits = ops.sink_hits
    m0, seen0, par0 = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "1"})
    assert ops.sink_hits == hits
    assert seen1 == seen0, (seen1, seen0)
    for k in par1:
        exact(par1[k], par0[k])
    # ... and with the projections' fused gather / activation epilogues (ops.tall_linear_act) switched off as well: the same
    # kernels' values through separate launches
    monkeypatch.setattr(ops, "TALL_FUSED", False)
    m2, seen2, par2 = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "1"})
    assert seen1 == seen2, (seen1, seen2)
    for k in par1:
        exact(par1[k], par2[k])


def test_captured_sampled_step_falls_back_on_a_batch_it_cannot_take(monkeypatch):
    """A pair whose interior plan is declined (here: forced) runs the ordinary eager step on its real shape, between two
    replays, on the same optimiser state -- and the epoch's numbers stay those of the all-eager loop to fp32 order."""
    from pygda_amd import sampled_graph as SG
    orig = SG._StaticBatch.takes
    calls = {"n": 0}

    def takes(self, slot, sizes):
        calls["n"] += 1
        return orig(self, slot, sizes) and calls["n"] != 7      # the 4th pair's source batch (two calls per accepted pair)
    monkeypatch.setattr(SG._StaticBatch, "takes", takes)
    mc, seen_c, par_c = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "1"}, dropout=0.0)
    st = mc._sampled_graphed[1]
    assert st.fallbacks == 1 and st.replays == 11, (st.fallbacks, st.replays)
    monkeypatch.setattr(SG._StaticBatch, "takes", orig)
    mr, seen_r, par_r = _sampled_fit(monkeypatch, {"PYGDA_AMD_SAMPLED_GRAPH": "0"}, dropout=0.0)
    np.testing.assert_allclose([v[0] for v in seen_c], [v[0] for v in seen_r], rtol=1e-5)
    for k in par_c:
        ref = par_r[k].double()                           # (in norm: see the captured == eager test above)
        rel = float((par_c[k].double() - ref).norm() / ref.norm().clamp_min(1e-30))
        assert rel <= 2e-2, (k, rel)


@pytest.mark.parametrize("p,pair", [(0.5, True), (0.3, False), (0.0, True)])
def test_interior_kstep_with_the_activation_in_its_epilogue(p, pair):
    """ops.propagate_act (gda_interior_kstep_lds_act_f32): the one-launch interior K-step writing dropout(relu(.)) itself
    -- twice with independent draws for ``pair`` -- against relu_dropout(propagate(.)) at the same dropout sites: outputs
    bit for bit (same pre-activation values, same keep-bits), and the gradients of x and the bias through autograd."""
    from pygda_amd.graph import as_graph
    n = 20000
    ei = _graph(n, 300000, 11, loops=True)
    g = torch.Generator().manual_seed(5)
    data = Data(x=torch.randn(n, 16, generator=g), edge_index=ei, y=torch.zeros(n, dtype=torch.long)).to(DEV)
    loader = NeighborLoader(data, [7, 5], batch_size=300, input_nodes=torch.randperm(n, generator=g)[:600], device=DEV)
    K, d = 10, 128
    st = ops.dropout_state
    for batch in loader:
        G = as_graph(batch.edge_index, batch.x.size(0))
        nb = batch.x.size(0)
        x = torch.randn(nb, d, generator=g).to(DEV)
        bias = torch.randn(d, generator=g).to(DEV)
        gy = torch.randn(nb, d, generator=g).to(DEV)
        assert ops.propagate_act_ok(x, G, K, bias)
        st.counter(x.device).fill_(3)
        xa, ba = x.clone().requires_grad_(), bias.clone().requires_grad_()
        st.site = 40
        got = ops.propagate_act(xa, G, K, ba, p, True, pair)
        g0 = got[0] if pair else got
        if pair:
            assert not got[1].requires_grad
        g0.backward(gy)
        xb, bb = x.clone().requires_grad_(), bias.clone().requires_grad_()
        st.site = 40
        pre = ops.propagate(xb, G, K, bb)
        want0 = ops.relu_dropout(pre, p, True)
        want1 = ops.relu_dropout(pre.detach(), p, True) if pair else None
        want0.backward(gy)
        exact(g0, want0)
        if pair:
            exact(got[1], want1)
            if p > 0:
                assert not torch.equal(got[0], got[1])                   # two draws, not one
        exact(xa.grad, xb.grad)
        exact(ba.grad, bb.grad)
        kept = float((g0 != 0).float().mean())
        assert 0.05 < kept < 0.6 * (1 - p) + 0.05                        # about half the pre-activations are positive
