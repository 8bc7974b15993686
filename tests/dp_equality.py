"""Harness for SURVEY 8(e)'s equality test: the W-rank averaged gradient of a data-parallel training
step equals the 1-rank gradient on the concatenated batch (disjoint union of the ranks' sampled
sub-graphs, CE means over all of their nodes, MMD over the concatenated row samples).

Trainers (``kind``): A2GNN with the MMD objective (``False`` / "a2gnn-mmd") and the adversarial one (``True`` /
"a2gnn-adv"); "udagcn" (pygda/models/udagcn.py:165-199: source CE + two gradient-reversed domain CEs + the target
entropy term, four node-count weighted means); "adagcn" (pygda/models/adagcn.py:169-198, 387-454: the critic loop --
Wasserstein gap of two global means + gradient penalty as a global mean over every rank's rows, critic gradients
averaged before each of its Adam steps -- then source CE + |gap|).  BASELINE.json configs[3] runs these two sharded
over ranks.

The ranks run ``pygda_amd``'s trainer code over a ``gloo`` group -- on the CPU (tests/
test_distributed_gloo.py, with the CPU oracle injected underneath the operator layer, because the
product kernels have no CPU path) or as two processes sharing one GPU (tests/test_gpu_configs.py,
the HIP kernels themselves).  The parent process then evaluates the concatenated batch without a
process group and compares.  Not a test module itself.
"""
import os
import socket

import torch

IN_DIM, HID, NCLS = 12, 8, 4
WEIGHT, ALPHA = 2.0, 0.3
TRAINER_KW = dict(num_layers=2, dropout=0.0, s_pnums=0, t_pnums=3, weight=WEIGHT, lr=0.01, epoch=1,
                  batch_size=40, num_neigh=[3, 2], verbose=0, use_hip_graph=False)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_domain(seed, n=150, e=600):
    from pygda_amd.data import Data
    g = torch.Generator().manual_seed(seed)
    return Data(x=torch.randn(n, IN_DIM, generator=g), edge_index=torch.randint(0, n, (2, e), generator=g),
                y=torch.randint(0, NCLS, (n,), generator=g))


def inject_oracle():
    """CPU runs only: put the CPU oracle underneath the operator layer (``propagate``, graph ingestion,
    row sampling, MMD on explicit rows), so that the trainer / loader / exchange-step logic of the
    product executes unchanged on CPU tensors."""
    from oracle import pygda_cpu as O
    import pygda_amd.nn.prop_gcn_conv as P
    import pygda_amd.utils.mmd as M

    def _graph(self, x, edge_index, edge_weight):
        return O.gcn_norm(edge_index, edge_weight, x.size(0))

    def propagate(x, graph, K=1, bias=None, colmajor_out=False):
        ei, w = graph
        for _ in range(K):
            x = O.propagate(ei, w, x)
        return x if bias is None else x + bias

    def mmd_rows(S, T):
        return sum(O.get_MMD(S[i], T[i], chunk_rows=250) for i in range(S.size(0))) / S.size(0)

    def grl_disc_ce(fs, ft, W, b, alpha, labels=None):          # a2gnn.py:197-205 on the oracle's GradReverse
        z = torch.nn.functional.linear(O.grad_reverse(torch.cat([fs, ft]), float(alpha)), W, b)
        y = torch.cat([torch.zeros(fs.size(0), dtype=torch.long), torch.ones(ft.size(0), dtype=torch.long)])
        return torch.nn.functional.cross_entropy(z, y)

    import pygda_amd.models.a2gnn as A
    import pygda_amd.nn.cached_gcn_conv as C

    def _cached_graph(self, x, edge_index, cache_name, edge_weight):        # cached_gcn_conv.py:88-103,132-136
        g = self.cache_dict.get(cache_name)
        if g is None:
            g = self.cache_dict[cache_name] = O.gcn_norm(edge_index, edge_weight, x.size(0), self.improved, True, "row")
        return g

    saved = [(A, "grl_disc_ce", A.grl_disc_ce), (P.PropGCNConv, "_graph", P.PropGCNConv._graph),
             (P, "propagate", P.propagate), (M, "sample_rows", M.sample_rows), (M, "mmd_loss_rows", M.mmd_loss_rows),
             (M, "mmd_loss", M.mmd_loss), (C.CachedGCNConv, "_graph", C.CachedGCNConv._graph),
             (C, "propagate", C.propagate)]
    A.grl_disc_ce = grl_disc_ce
    P.PropGCNConv._graph = _graph
    P.propagate = propagate
    C.CachedGCNConv._graph = _cached_graph
    C.propagate = propagate
    M.sample_rows = lambda feat, idx, sel=None: feat[idx]
    M.mmd_loss_rows = mmd_rows
    M.mmd_loss = lambda sf, tf, si, ti, *a, **k: mmd_rows(sf[si], tf[ti])

    def restore():               # the parent (pytest) process goes on to other tests: the product layer back in place
        for obj, name, val in saved:
            setattr(obj, name, val)
    return restore


def _kind(adv):
    """Back-compatible spelling: ``adv`` False / True = A2GNN with the MMD / adversarial objective."""
    return {False: "a2gnn-mmd", True: "a2gnn-adv"}.get(adv, adv)


def _no_dropout(*modules):
    for module in modules:
        for m in module.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            for d in getattr(m, "dropout_layers", []):       # UDAGCN: unregistered Dropout(0.1) modules
                d.p = 0.0


UDA_ALPHA, UDA_EPOCH = 0.05, 7


def _trainer(device, adv):
    import pygda_amd
    kind = _kind(adv)
    if kind == "udagcn":       # no PPMI view: its random-walk graph of a union batch is a different draw by construction
        kw = {k: v for k, v in TRAINER_KW.items() if k not in ("s_pnums", "t_pnums", "weight", "dropout", "epoch")}
        return pygda_amd.models.UDAGCN(IN_DIM, HID, NCLS, ppmi=False, adv_dim=6, device=device, epoch=20, **kw)
    if kind == "adagcn":
        kw = {k: v for k, v in TRAINER_KW.items() if k not in ("s_pnums", "t_pnums", "weight", "dropout")}
        return pygda_amd.models.AdaGCN(IN_DIM, HID, NCLS, adv_dim=6, gp_weight=5, domain_weight=1, weight_decay=0.01,
                                       device=device, **kw)
    return pygda_amd.models.A2GNN(IN_DIM, HID, NCLS, adv=kind == "a2gnn-adv", device=device, **TRAINER_KW)


def _net_of(m):
    return getattr(m, "a2gnn", None) or getattr(m, "udagcn", None) or getattr(m, "adagcn", None)


def _np(t):
    return t.detach().cpu().numpy().copy()        # numpy payloads: torch tensors would travel as shared-memory
                                                   # handles that die with the worker process


def _cpu(batch):
    return dict(x=_np(batch.x), edge_index=_np(batch.edge_index), y=_np(batch.y))


def _tensors(d):
    return {k: (torch.from_numpy(v) if hasattr(v, "dtype") and not torch.is_tensor(v) else v) for k, v in d.items()}


def _step(m, kind, src, tgt):
    if kind == "udagcn":
        return m.forward_model(src, tgt, UDA_ALPHA, UDA_EPOCH)
    if kind == "adagcn":
        return m.forward_model(src, tgt)
    return m.forward_model(src, tgt, ALPHA)


def worker(rank, world, port, q, device, adv, oracle):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYGDA_AMD_HIPGRAPH="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if oracle:
            inject_oracle()
        kind = _kind(adv)
        from pygda_amd import distributed as D
        torch.manual_seed(7 + rank)                      # ranks initialise differently on purpose
        m = _trainer(device, adv)
        prepared = m._prepare(make_domain(1), make_domain(2, n=130, e=500))
        net, optimizer = prepared[0], prepared[1]
        aux = list(getattr(m, "_dp_aux_modules", ()))    # AdaGCN's critic (own optimiser)
        _no_dropout(net, *aux)
        D.broadcast_parameters(net)                      # what _train_epochs does at the first epoch
        for a in aux:
            D.broadcast_parameters(a)
        extra = {}
        if kind == "adagcn":
            extra["disc0"] = {k: _np(v) for k, v in m.discriminator.state_dict().items()}
        if len(prepared) > 4 and prepared[4] is not None:
            prepared[4]()                                # before_step: UDAGCN switches its model list to train()
        else:
            net.train()
        src = next(iter(m.source_loader)).to(device)
        tgt = next(iter(m.target_loader)).to(device)
        torch.manual_seed(100 + rank)                    # this rank's host draws: MMD rows / interpolation weights
        loss, _, _ = _step(m, kind, src, tgt)
        optimizer.zero_grad()
        loss.backward()
        D.allreduce_grads(p for g in optimizer.param_groups for p in g["params"])
        if kind == "adagcn":
            extra["disc10"] = {k: _np(v) for k, v in m.discriminator.state_dict().items()}
        q.put((rank, dict(state={k: _np(v) for k, v in net.state_dict().items()},
                          grads={k: _np(p.grad) for k, p in net.named_parameters() if p.grad is not None},
                          src=_cpu(src), tgt=_cpu(tgt), loss=float(loss.detach()), **extra)))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def run_ranks(world, device, adv, oracle, timeout=600):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, device, adv, oracle)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=timeout) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r, v in got.items():
        assert isinstance(v, dict), f"rank {r}:\n{v}"
    out = []
    for r in range(world):
        d = dict(state=_tensors(got[r]["state"]), grads=_tensors(got[r]["grads"]), src=_tensors(got[r]["src"]),
                 tgt=_tensors(got[r]["tgt"]), loss=got[r]["loss"])
        for k in ("disc0", "disc10"):
            if k in got[r]:
                d[k] = _tensors(got[r][k])
        out.append(d)
    return out


def concatenated_reference(results, device, adv, oracle):
    """The same objective on ONE process: union batch, concatenated row samples."""
    from pygda_amd.data import Data
    restore = inject_oracle() if oracle else None
    try:
        return _concatenated_reference(results, device, adv, Data)
    finally:
        if restore is not None:
            restore()


def _adagcn_union_step(m, src, tgt, ns, nt, device):
    """AdaGCN's step (adagcn.py:138-198) on the union batch as ONE process evaluates it: every mean is over all the
    rows of the union.  The gradient penalty's interpolates pair source and target rows by position (:423-434);
    the data-parallel step pairs them inside each rank's own batch and draws each rank's weights from that rank's
    generator, so the union objective is formed over exactly those rows: per rank block, the reference's own pairing
    rule and that rank's draws, all rows under one mean."""
    net, disc = m.adagcn, m.discriminator
    world = len(ns)
    so = [sum(ns[:r]) for r in range(world + 1)]
    to = [sum(nt[:r]) for r in range(world + 1)]
    gens = []
    for r in range(world):                                  # each rank's CPU generator, advanced in its own order
        g = torch.Generator()
        g.manual_seed(100 + r)
        gens.append(g)
    h0_s, h0_t = net.first_conv(src), net.first_conv(tgt)
    for _ in range(m.critic_steps):
        with torch.no_grad():
            es, et = net.forward_from(h0_s.detach(), src), net.forward_from(h0_t.detach(), tgt)
        inter = []
        for r in range(world):
            a, b = es[so[r]:so[r + 1]], et[to[r]:to[r + 1]]
            k = min(a.size(0), b.size(0))
            if a.size(0) < b.size(0):
                hs, ht = torch.cat((a, a)), torch.cat((b[0:k], b[-k:]))
            elif a.size(0) > b.size(0):
                hs, ht = torch.cat((a[0:k], a[-k:])), torch.cat((b, b))
            else:
                hs, ht = a, b
            al = torch.rand((hs.size(0), 1), generator=gens[r]).to(device)
            inter.append(ht + al * (hs - ht))
        inputs = torch.cat([es, et] + inter).requires_grad_(True)
        scores = disc(inputs)
        grad = torch.autograd.grad(inputs=inputs, outputs=scores, grad_outputs=torch.ones_like(scores),
                                   create_graph=True, retain_graph=True, only_inputs=True)[0]
        gp = torch.mean((grad.view(grad.shape[0], -1).norm(2, dim=1) - 1) ** 2)
        gap = torch.mean(disc(es).reshape(-1)) - torch.mean(disc(et).reshape(-1))
        m.c_optimizer.zero_grad()
        (-torch.abs(gap) + m.gp_weight * gp).backward()
        m.c_optimizer.step()
    es, et = net.forward_from(h0_s, src), net.forward_from(h0_t, tgt)
    logits = net.cls_model(es)
    gap = torch.mean(disc(es).reshape(-1)) - torch.mean(disc(et).reshape(-1))
    return net.loss_func(logits, src.y) + torch.abs(gap) * m.domain_weight


def _concatenated_reference(results, device, adv, Data):
    world = len(results)
    kind = _kind(adv)

    def union(key):
        xs, eis, ys, off, sizes = [], [], [], 0, []
        for r in results:
            b = r[key]
            xs.append(b["x"]); ys.append(b["y"]); eis.append(b["edge_index"] + off)
            sizes.append(b["x"].size(0)); off += b["x"].size(0)
        return Data(x=torch.cat(xs), edge_index=torch.cat(eis, dim=1), y=torch.cat(ys)).to(device), sizes

    src, ns = union("src")
    tgt, nt = union("tgt")
    m = _trainer(device, adv)
    if kind in ("udagcn", "adagcn"):
        net = m.init_model().to(device)
        setattr(m, kind, net)
        net.load_state_dict(results[0]["state"])
        _no_dropout(net)
        net.train()
        net.zero_grad()
        if kind == "udagcn":
            for mod in net.models:
                mod.train()
            loss, _, _ = m.forward_model(src, tgt, UDA_ALPHA, UDA_EPOCH)
        else:
            m.discriminator = torch.nn.Sequential(torch.nn.Linear(HID, 6), torch.nn.ReLU(), torch.nn.Dropout(0.0),
                                                  torch.nn.Linear(6, 1), torch.nn.Sigmoid()).to(device)
            m.discriminator.load_state_dict(results[0]["disc0"])
            m.discriminator.train()
            m.c_optimizer = torch.optim.Adam(m.discriminator.parameters(), lr=m.lr, weight_decay=m.weight_decay)
            loss = _adagcn_union_step(m, src, tgt, ns, nt, device)
        loss.backward()
        extra = {}
        if kind == "adagcn":
            extra["disc10"] = {k: v.detach().cpu() for k, v in m.discriminator.state_dict().items()}
        grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
        return float(loss.detach()), grads, (ns, nt, extra)
    m.a2gnn = m.init_model().to(device)
    m.a2gnn.load_state_dict(results[0]["state"])
    m.a2gnn.train()
    m.a2gnn.zero_grad()
    if kind == "a2gnn-adv":
        loss, _, _ = m.forward_model(src, tgt, ALPHA)
    else:
        ce, _, sf, tf, _, pending, _ = m._branches(src, tgt)
        if pending is not None:
            torch.cuda.current_stream().wait_stream(pending[1])
        times, per = 5, -(-1000 // world)
        idx_s, idx_t, off_s, off_t = [], [], 0, 0
        for r in range(world):                            # each rank's draws, in its generator order
            torch.manual_seed(100 + r)
            idx_s.append(torch.randint(ns[r], (times, per)) + off_s)
            idx_t.append(torch.randint(nt[r], (times, per)) + off_t)
            off_s += ns[r]; off_t += nt[r]
        idx_s, idx_t = torch.cat(idx_s, dim=1).to(device), torch.cat(idx_t, dim=1).to(device)
        import pygda_amd.utils.mmd as M
        loss = ce + WEIGHT * M.mmd_loss(sf, tf, idx_s, idx_t)
    loss.backward()
    return float(loss.detach()), {k: p.grad.detach().cpu() for k, p in m.a2gnn.named_parameters()}, (ns, nt)
