"""GPU: the headline configuration at its FULL size against the CPU oracle ("what's weak" 1-3).

* cfg-A (BASELINE.json configs[1]: A2GNN ACMv9 -> DBLPv7 shapes, F = 6,775, nhid = 128, L = 2, t_pnums = 10) as a
  TRAINING STEP: loss, both logits, predicted labels and every parameter gradient of one ``forward_model`` +
  backward with the seeded 5 x 1000 MMD draw (pygda/models/a2gnn.py:146-213, pygda/utils/mmd.py:148-149), dropout 0,
  against ``oracle.a2gnn_forward_model`` -- on the uniform stand-in graphs and on the power-law ones (hub rows of
  several hundred entries: the regime of real citation graphs).  The one-launch K-step kernel, the sparse layer 0,
  the stacked source pass and the fused MMD are exercised together here at the size the benchmark runs them.
* three epochs of ``fit()`` at that size: the hipGraph-captured loop against eager launches, and the eager loop
  against the oracle's loop (same seeds, same CPU-generator draws, torch's Adam on both sides).
* configs[4] at its full per-domain size (5 M nodes / 105 M entries, d = 128): size-independent properties of the
  aggregation operator (adjoint identity, K steps == K single steps bit for bit, row sums, linearity).

Tolerances: 1e-4 absolute on logits, 1e-4 relative on losses (BASELINE.json north_star), labels identical;
gradients per tensor: relative L2 error <= 1e-4 AND the element-wise bound of tests/test_gpu_parity.py.
"""
import numpy as np
import psutil
import pytest
import torch

import pygda_amd
from pygda_amd import ops
from pygda_amd.graph import build_csr
from oracle import pygda_cpu as O
from tests.test_gpu_parity import DEV, LOGIT_ATOL, REL, close, exact

pytestmark = pytest.mark.gpu

HP = dict(num_layers=2, s_pnums=0, t_pnums=10, weight=10, lr=0.01, weight_decay=0.005)


def _mmd_chunk():
    """The oracle's [2000, 2000, 128] temporaries take ~20 GB at the reference's own evaluation order; on a host
    with less memory they are formed 128 rows at a time (same per-element arithmetic, oracle/pygda_cpu.py)."""
    return None if psutil.virtual_memory().available > 48 * 2 ** 30 else 128


def _rel_l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def _check_grads(net, ora, rel=1e-4, elem=1e-4):
    """Every parameter gradient of the product against the oracle's: relative L2 per tensor, the element-wise bound of
    tests/test_gpu_parity.py, and sign agreement of every entry that is not noise-sized (a sign error in a small
    gradient hides in an absolute bound).  Returns the number of tensors compared."""
    named = dict(ora.named_parameters())
    checked = 0
    top = max(float(q.grad.abs().max()) for q in named.values() if q.grad is not None)
    for k, p in net.named_parameters():
        v = named[k].grad
        if v is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        if float(v.abs().max()) <= 1e-7 * top:
            # a gradient that is zero in exact arithmetic (the bias of UDAGCN's view attention: softmax over the views
            # ignores a shift common to both) is rounding residue on both sides, ~1e-11 here: nothing to compare
            assert float(p.grad.abs().max()) <= 1e-6 * top, k
            checked += 1
            continue
        assert _rel_l2(p.grad, v) <= rel, (k, _rel_l2(p.grad, v))
        close(p.grad, v, rtol=1e-3, atol=elem * max(float(v.abs().max()), 1e-3))
        big = v.abs() > max(1e-3, 10 * elem) * float(v.abs().max())
        assert bool((torch.sign(p.grad.detach().cpu()[big]) == torch.sign(v[big])).all()), k
        checked += 1
    return checked


def _model(feat, dropout=0.0, epoch=3, **kw):
    return pygda_amd.models.A2GNN(feat, 128, 5, dropout=dropout, device=DEV, epoch=epoch, verbose=0, **HP, **kw)


@pytest.mark.parametrize("graph", ["uniform", "powerlaw"])
def test_cfg_a_full_size_training_step_vs_oracle(graph):
    from bench import make_cfg_a
    src, tgt = make_cfg_a(seed=200, degrees=graph)
    if graph == "powerlaw":            # hubs far beyond what one lane of the K-step kernel holds
        deg = torch.bincount(tgt.edge_index[1], minlength=tgt.num_nodes)
        assert int(deg.max()) >= 300
    m = _model(src.x.size(1))
    torch.manual_seed(11)
    m.a2gnn = m.init_model()
    ora = O.A2GNNBase(src.x.size(1), 128, 5, num_layers=2, dropout=0.0)
    ora.load_state_dict({k: v.detach().cpu() for k, v in m.a2gnn.state_dict().items()})
    m.a2gnn.train(); ora.train()
    torch.manual_seed(77)              # the 5 x 1000 row draws MMD() takes from the CPU generator
    loss, sl, tl = m.forward_model(src.to(DEV), tgt.to(DEV), 0.3)
    loss.backward()
    torch.manual_seed(77)
    want, wsl, wtl = O.a2gnn_forward_model(ora, O.Graph(src.x, src.edge_index, src.y),
                                           O.Graph(tgt.x, tgt.edge_index, tgt.y), 0.3, HP["s_pnums"], HP["t_pnums"],
                                           False, HP["weight"], _mmd_chunk())
    want.backward()
    close(loss, want, rtol=REL)
    close(sl, wsl, rtol=0, atol=LOGIT_ATOL)
    close(tl, wtl, rtol=0, atol=LOGIT_ATOL)
    exact(sl.argmax(1), wsl.argmax(1))
    exact(tl.argmax(1), wtl.argmax(1))
    assert _check_grads(m.a2gnn, ora) >= 6                # 3 convs x (weight, bias)


def _fit(src, tgt, use_hip_graph, seed=5):
    m = _model(src.x.size(1), use_hip_graph=use_hip_graph)
    seen = []
    m.epoch_hook = lambda e, loss, acc, secs: seen.append((loss, acc))
    torch.manual_seed(seed)
    m.fit(src, tgt)
    logits, labels = m.predict(tgt)
    return m, seen, logits, labels


def test_cfg_a_full_size_fit_captured_vs_eager_vs_oracle(monkeypatch):
    from bench import make_cfg_a
    import os
    src, tgt = make_cfg_a(seed=int(os.environ.get("FULLSIZE_SEED", "200")))
    m_e, seen_e, logits_e, labels_e = _fit(src, tgt, False)
    monkeypatch.setenv("PYGDA_AMD_GRAPH_UNROLL", "2")        # three epochs = a two-step replay + a one-step replay
    m_c, seen_c, logits_c, _ = _fit(src, tgt, True)
    assert getattr(m_c, "_graphed", None) is not None, "the captured path did not run"
    assert len(seen_e) == len(seen_c) == 3
    close([s[0] for s in seen_c], [s[0] for s in seen_e], rtol=REL)
    close([s[1] for s in seen_c], [s[1] for s in seen_e], rtol=0, atol=1e-12)
    close(logits_c, logits_e, rtol=0, atol=LOGIT_ATOL)
    exact(logits_c.argmax(1), logits_e.argmax(1))
    # the oracle's loop (a2gnn.py:298-336): same init stream, same MMD draws, torch.optim.Adam
    torch.manual_seed(5)
    ora = O.A2GNNBase(src.x.size(1), 128, 5, num_layers=2, dropout=0.0)
    opt = torch.optim.Adam(ora.parameters(), lr=HP["lr"], weight_decay=HP["weight_decay"])
    s, t = O.Graph(src.x, src.edge_index, src.y), O.Graph(tgt.x, tgt.edge_index, tgt.y)
    losses, accs = [], []
    for _ in range(3):
        val, s_logits = O.a2gnn_train_step(ora, opt, s, t, 0.0, HP["s_pnums"], HP["t_pnums"], False, HP["weight"],
                                           _mmd_chunk())
        losses.append(val)
        accs.append(float((s_logits.argmax(1) == src.y).float().mean()))
    ora.eval()
    with torch.no_grad():
        want = ora(t, HP["t_pnums"])
    close([s[0] for s in seen_e], losses, rtol=REL)
    close([s[1] for s in seen_e], accs, rtol=0, atol=2.0 / src.num_nodes)      # a near-tie row may flip
    # After three Adam steps the 1e-4 bound of a single pass no longer applies to ANY two fp32 executions of the
    # reference's loop: Adam's first step moves every weight by lr * sign(g), so an entry of the 867 k-element layer-0
    # weight whose gradient is summation-order noise lands 2 lr = 0.02 away when its sign comes out differently, and ten
    # propagation steps spread that over the graph.  Measured: the CPU oracle against itself with 8 vs 3 threads ends
    # 2.8e-4 apart on these logits; on the GPU (round 4, two stand-in seeds x both MMD kernel paths) the largest
    # deviation from the oracle was 4.2e-4 ... 7.3e-4, while the SHARE of logits beyond 1e-4 is a lottery on how many
    # such entries there are -- seed 200: 0.14 % (two-pass MMD) / 4.7 % (one-pass), seed 201: 54.8 % / 54.6 % -- and
    # is printed, not asserted.  So: 1e-3 absolute on every logit and labels agreeing on 99.9 % of the nodes; the per-epoch
    # losses above and the single training step (test_cfg_a_full_size_training_step_vs_oracle) keep the 1e-4 bounds.
    close(logits_e, want, rtol=0, atol=1e-3)
    dev = (logits_e.cpu() - want).abs()
    print(f"logits beyond {LOGIT_ATOL} of the oracle's after three Adam steps: {float((dev > LOGIT_ATOL).float().mean()):.4f}, "
          f"largest deviation {float(dev.max()):.2e}")
    agree = float((logits_e.argmax(1).cpu() == want.argmax(1)).float().mean())
    assert agree >= 0.999, agree
    exact(labels_e, tgt.y)


def test_cfg_a_one_pass_mmd_is_not_worse_than_two_pass_after_training(monkeypatch):
    """the one-pass MMD (split-fp16 MFMAs, csrc/gda_mmd_fused.inc) against the two-pass fp32-MFMA
    kernels AFTER training, over three stand-in seeds: three Adam steps of the full-size cfg-A fit with each kernel path,
    both measured against the CPU oracle's loop on the same inputs (same init stream, same MMD draws).  Round 4 only
    PRINTED the share of logits beyond 1e-4 (seed 200: 4.7 % one-pass against 0.14 % two-pass; seed 201: 54.6 % / 54.8 %)
    and called it a lottery on which near-zero gradient entries change sign under Adam's sign-like first steps.  If that
    is what it is, the one-pass path must not be SYSTEMATICALLY further from the oracle: averaged over the seeds its
    largest logit deviation and its RMS deviation stay within 1.5 x the two-pass path's (a floor of 5e-5 / 1e-5 keeps a
    near-perfect two-pass seed from turning the ratio into noise), no single seed beyond 3 x, and both paths keep the
    1e-3 bound of the trajectory test."""
    from bench import make_cfg_a
    rows = []
    for seed in (200, 201, 202):
        src, tgt = make_cfg_a(seed=seed)
        torch.manual_seed(5)
        ora = O.A2GNNBase(src.x.size(1), 128, 5, num_layers=2, dropout=0.0)
        opt = torch.optim.Adam(ora.parameters(), lr=HP["lr"], weight_decay=HP["weight_decay"])
        s, t = O.Graph(src.x, src.edge_index, src.y), O.Graph(tgt.x, tgt.edge_index, tgt.y)
        for _ in range(3):
            O.a2gnn_train_step(ora, opt, s, t, 0.0, HP["s_pnums"], HP["t_pnums"], False, HP["weight"], _mmd_chunk())
        ora.eval()
        with torch.no_grad():
            want = ora(t, HP["t_pnums"])
        dev = {}
        for one_pass in (True, False):
            monkeypatch.setattr(ops, "MMD_ONE_PASS", one_pass)
            _, _, logits, _ = _fit(src, tgt, False)
            d = (logits.cpu() - want).abs()
            dev[one_pass] = (float(d.max()), float(d.pow(2).mean().sqrt()), float((d > LOGIT_ATOL).float().mean()))
            assert dev[one_pass][0] <= 1e-3, (seed, one_pass, dev[one_pass])
        monkeypatch.undo()
        rows.append((seed, dev[True], dev[False]))
        print(f"seed {seed}: one-pass max {dev[True][0]:.2e} rms {dev[True][1]:.2e} share>1e-4 {dev[True][2]:.4f} | "
              f"two-pass max {dev[False][0]:.2e} rms {dev[False][1]:.2e} share>1e-4 {dev[False][2]:.4f}")
        assert dev[True][0] <= 3.0 * max(dev[False][0], 5e-5), rows[-1]
    mean = lambda k, which: sum(r[which][k] for r in rows) / len(rows)
    assert mean(0, 1) <= 1.5 * max(mean(0, 2), 5e-5), rows             # largest deviation, averaged over the seeds
    assert mean(1, 1) <= 1.5 * max(mean(1, 2), 1e-5), rows             # RMS deviation, averaged over the seeds


def test_cfg_s_full_size_aggregation_properties():
    """configs[4]: one domain at its full size -- 5 M nodes, 100 M directed edges + self loops, d = 128."""
    n, deg, d = 5_000_000, 20, 128
    gen = torch.Generator(device=DEV).manual_seed(200)
    half = n * deg // 2
    a = torch.randint(0, n, (half,), generator=gen, device=DEV)
    b = torch.randint(0, n, (half,), generator=gen, device=DEV)
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    loops = int((a == b).sum())
    del a, b
    G = build_csr(ei, n, validate=False)
    assert G.nnz == n * deg + n - 2 * loops
    x = torch.randn(n, d, generator=gen, device=DEV)
    y = torch.randn(n, d, generator=gen, device=DEV)
    ax = ops.spmm_kstep(G, x, 1)
    aty = ops.spmm_kstep(G, y, 1, None, transposed=True)
    lhs, rhs = float((ax.double() * y.double()).sum()), float((x.double() * aty.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0)
    del aty
    exact(ops.spmm_kstep(G, x, 2), ops.spmm_kstep(G, ax, 1))             # K steps == K single steps, bit for bit
    lin = ops.spmm_kstep(G, 2.0 * x + y, 1)
    close(lin[:200_000], (2.0 * ax + ops.spmm_kstep(G, y, 1))[:200_000], rtol=1e-5, atol=1e-5)
    del lin, x, y, ax
    rowsum = ops.spmm_kstep(G, torch.ones(n, 4, device=DEV), 1)[:, 0]
    w = torch.zeros(n, device=DEV, dtype=torch.float64).index_add_(
        0, torch.repeat_interleave(torch.arange(n, device=DEV), (G.rowptr[1:] - G.rowptr[:-1]).long()),
        G.val[:G.nnz].double())
    close(rowsum, w, rtol=1e-5, atol=1e-6)
    # the symmetric graph's normalised operator is symmetric: forward and transposed CSR give the same product
    z = torch.randn(n, 8, generator=gen, device=DEV)
    close(ops.spmm_kstep(G, z, 1), ops.spmm_kstep(G, z, 1, None, transposed=True), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: UDAGCN / AdaGCN, ACMv9 -> Citationv1 (benchmark/node/run_citation.sh:109-110: nhid 128, num_layers 2;
# UDAGCN lr 1e-4, weight_decay 1e-3, 400 epochs; AdaGCN lr 0.01, weight_decay 0.01, 400 epochs) at the datasets' own
# sizes (Ns = 9,360 / Es = 15,556, Nt = 8,935 / Et = 15,098 undirected pairs, F = 6,775), dropout 0, one training step
# against the oracle: loss 1e-4 relative, logits 1e-4 absolute, labels identical, every parameter gradient.
# ---------------------------------------------------------------------------------------------------------------------
def _cfg_c():
    from bench import make_cfg_a
    return make_cfg_a(seed=203, ns=9360, es=15556, nt=8935, et=15098)


def _no_dropout(*modules):
    for module in modules:
        for m in module.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            for d in getattr(m, "dropout_layers", []):     # UDAGCN keeps its Dropout(0.1) modules in a plain list
                d.p = 0.0


@pytest.mark.parametrize("ppmi", [False, True])
def test_udagcn_full_size_step_vs_oracle(ppmi, monkeypatch):
    """pygda/models/udagcn.py:131-201 at the size of configs[3].  With the PPMI view the random-walk graphs are the
    ones the DEVICE builder produced in this very forward pass (the reference's np.random walk stream cannot be
    replayed): their RAW weighted edge lists are recorded on the way and handed to the oracle, which adds the self
    loops and applies the source-degree normalisation itself (ppmi_conv.py:171-184, float64) -- so the ingestion of
    the PPMI graph, both aggregation stacks on shared weights, the attention fusion, the fused GRL + two-layer
    discriminator + CE kernels and the entropy term are all on the checked path."""
    src, tgt = _cfg_c()
    feat = src.x.size(1)
    m = pygda_amd.models.UDAGCN(feat, 128, 5, num_layers=2, ppmi=ppmi, adv_dim=40, lr=1e-4, weight_decay=1e-3,
                                epoch=400, device=DEV, verbose=0, use_hip_graph=False)
    recorded = []
    if ppmi:
        import pygda_amd.nn.ppmi_conv as PC
        real = PC.ppmi_edges

        def spy(edge_index, num_nodes, path_len=5, **kw):
            ei, w = real(edge_index, num_nodes, path_len, **kw)
            recorded.append((ei.detach().cpu(), w.detach().cpu(), int(path_len)))
            return ei, w
        monkeypatch.setattr(PC, "ppmi_edges", spy)
    torch.manual_seed(11); np.random.seed(11)
    m.udagcn = m.init_model()
    _no_dropout(m.udagcn)
    for mod in m.udagcn.models:
        mod.train()
    ora = O.UDAGCNBase(feat, 128, 5, num_layers=2, ppmi=ppmi, adv_dim=40, dropout_p=0.0)
    ora.load_state_dict({k: v.detach().cpu() for k, v in m.udagcn.state_dict().items()})
    ora.train()
    alpha, epoch = 0.05, 37                       # the schedule's value from epoch 19 on (udagcn.py: min((e+1)/epochs, 0.05))
    loss, sl, tl = m.forward_model(src.to(DEV), tgt.to(DEV), alpha, epoch)
    loss.backward()
    if ppmi:
        # call order of the product's forward: source layers 0..L-1, then target layers 0..L-1 (each PPMIConv walks
        # its own graph, ppmi_conv.py:132-136 caches per layer)
        assert len(recorded) == 4 and all(r[2] == 10 for r in recorded)
        it = iter(recorded)
        for name, n in (("source", src.num_nodes), ("target", tgt.num_nodes)):
            for conv in ora.ppmi_encoder.conv_layers:
                ei, w, _ = next(it)
                assert ei.size(1) > 4 * n                 # a real random-walk graph, not a degenerate one
                ei2, w2 = O.add_remaining_self_loops(ei, w.double(), 1, n)
                row, col = ei2
                deg = torch.zeros(n, dtype=torch.float64).index_add_(0, row, w2)
                dis = deg.pow(-0.5)
                dis[dis == float("inf")] = 0
                conv.cache_dict[name] = (ei2, (dis[row] * w2 * dis[col]).float())
    want, wsl, wtl = O.udagcn_forward_model(ora, O.Graph(src.x, src.edge_index, src.y),
                                            O.Graph(tgt.x, tgt.edge_index, tgt.y), alpha, epoch, 400)
    want.backward()
    close(loss, want, rtol=REL)
    close(sl, wsl, rtol=0, atol=LOGIT_ATOL)
    close(tl, wtl, rtol=0, atol=LOGIT_ATOL)
    exact(sl.argmax(1), wsl.argmax(1))
    exact(tl.argmax(1), wtl.argmax(1))
    # encoder convs (shared by both views) 2 x (weight, bias), classifier 2, discriminator 4 (+ attention 2 with ppmi)
    assert _check_grads(m.udagcn, ora) >= (12 if ppmi else 10)


def test_adagcn_full_size_step_vs_oracle():
    """pygda/models/adagcn.py:138-198, 387-454 at the size of configs[3], in the two stages the step consists of.

    (1) The whole ``forward_model``: ten critic updates (Wasserstein gap + gradient penalty on CPU-generator
    interpolation weights, Adam on the critic), then the encoder loss.  Loss and logits hold the 1e-4 bounds.  The
    critic after ten Adam steps is compared at relative L2 1e-3 and the encoder gradients of this stage at 5e-3:
    Adam's first steps move every weight by ~lr * sign(gradient), so entries whose gradient is summation-order noise
    end up to 2 lr apart between ANY two fp32 evaluations (measured on the first run of this test: critic within 1e-3,
    the encoder gradient -- linear in the critic's weights through |E D(s) - E D(t)| -- 1.3e-3 apart on the layer-0
    bias while every tensor of the critic-free UDAGCN step above held 1e-4).
    (2) The encoder objective GIVEN the critic: the oracle's ten-step critic is loaded into the product's and both
    evaluate the encoder loss with no further critic update (``critic_steps = 0``): loss, logits, labels and every
    parameter gradient at the 1e-4 bounds of this file."""
    src, tgt = _cfg_c()
    feat = src.x.size(1)
    m = pygda_amd.models.AdaGCN(feat, 128, 5, num_layers=2, adv_dim=40, gp_weight=5, domain_weight=1, lr=0.01,
                                weight_decay=0.01, epoch=400, device=DEV, verbose=0, use_hip_graph=False)
    torch.manual_seed(11)
    net, _, _, _ = m._prepare(src, tgt)            # loaders, encoder, the critic and its optimiser (adagcn.py:254-275)
    _no_dropout(net, m.discriminator)
    net.train(); m.discriminator.train()
    ora = O.AdaGCNBase(feat, 128, 5, num_layers=2, dropout_p=0.0)
    ora.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    ora.train()
    odisc = torch.nn.Sequential(torch.nn.Linear(128, 40), torch.nn.ReLU(), torch.nn.Dropout(0.0),
                                torch.nn.Linear(40, 1), torch.nn.Sigmoid())
    odisc.load_state_dict({k: v.detach().cpu() for k, v in m.discriminator.state_dict().items()})
    c_opt = torch.optim.Adam(odisc.parameters(), lr=0.01, weight_decay=0.01)
    (s,), (t,) = list(m.source_loader), list(m.target_loader)
    s, t = s.to(DEV), t.to(DEV)
    Gs, Gt = O.Graph(src.x, src.edge_index, src.y), O.Graph(tgt.x, tgt.edge_index, tgt.y)
    # ---- stage 1: the whole step
    torch.manual_seed(77)                          # the interpolation weights of the ten gradient penalties
    loss, sl, tl = m.forward_model(s, t)
    net.zero_grad()
    loss.backward()
    torch.manual_seed(77)
    want, wsl, wtl = O.adagcn_forward_model(ora, odisc, c_opt, Gs, Gt, 5, 1, 10)
    ora.zero_grad()
    want.backward()
    before = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for k, v in odisc.state_dict().items():
        got = m.discriminator.state_dict()[k]
        assert _rel_l2(got, v) <= 1e-3, (k, _rel_l2(got, v))
    close(loss, want, rtol=REL)
    close(sl, wsl, rtol=0, atol=LOGIT_ATOL)
    close(tl, wtl, rtol=0, atol=LOGIT_ATOL)
    exact(sl.argmax(1), wsl.argmax(1))
    exact(tl.argmax(1), wtl.argmax(1))
    assert _check_grads(net, ora, rel=5e-3, elem=5e-3) >= 6   # 2 convs x (weight, bias) + classifier
    for k, v in net.state_dict().items():          # forward_model trains the critic only
        exact(v, before[k])
    # ---- stage 2: the encoder objective given the (oracle's) critic
    m.discriminator.load_state_dict({k: v.to(DEV) for k, v in odisc.state_dict().items()})
    m.critic_steps = 0
    loss, sl, tl = m.forward_model(s, t)
    net.zero_grad()
    loss.backward()
    want, wsl, wtl = O.adagcn_forward_model(ora, odisc, c_opt, Gs, Gt, 5, 1, 0)
    ora.zero_grad()
    want.backward()
    close(loss, want, rtol=REL)
    close(sl, wsl, rtol=0, atol=LOGIT_ATOL)
    close(tl, wtl, rtol=0, atol=LOGIT_ATOL)
    exact(sl.argmax(1), wsl.argmax(1))
    exact(tl.argmax(1), wtl.argmax(1))
    assert _check_grads(net, ora) >= 6
