"""Harness for SURVEY 8(e)'s equality test: the W-rank averaged gradient of a data-parallel A2GNN
step equals the 1-rank gradient on the concatenated batch (disjoint union of the ranks' sampled
sub-graphs, CE means over all of their nodes, MMD over the concatenated row samples).

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
    saved = [(A, "grl_disc_ce", A.grl_disc_ce), (P.PropGCNConv, "_graph", P.PropGCNConv._graph),
             (P, "propagate", P.propagate), (M, "sample_rows", M.sample_rows), (M, "mmd_loss_rows", M.mmd_loss_rows),
             (M, "mmd_loss", M.mmd_loss)]
    A.grl_disc_ce = grl_disc_ce
    P.PropGCNConv._graph = _graph
    P.propagate = propagate
    M.sample_rows = lambda feat, idx, sel=None: feat[idx]
    M.mmd_loss_rows = mmd_rows
    M.mmd_loss = lambda sf, tf, si, ti, *a, **k: mmd_rows(sf[si], tf[ti])

    def restore():               # the parent (pytest) process goes on to other tests: the product layer back in place
        for obj, name, val in saved:
            setattr(obj, name, val)
    return restore


def _trainer(device, adv):
    import pygda_amd
    return pygda_amd.models.A2GNN(IN_DIM, HID, NCLS, adv=adv, device=device, **TRAINER_KW)


def _np(t):
    return t.detach().cpu().numpy().copy()        # numpy payloads: torch tensors would travel as shared-memory
                                                   # handles that die with the worker process


def _cpu(batch):
    return dict(x=_np(batch.x), edge_index=_np(batch.edge_index), y=_np(batch.y))


def _tensors(d):
    return {k: (torch.from_numpy(v) if hasattr(v, "dtype") and not torch.is_tensor(v) else v) for k, v in d.items()}


def worker(rank, world, port, q, device, adv, oracle):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYGDA_AMD_HIPGRAPH="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if oracle:
            inject_oracle()
        from pygda_amd import distributed as D
        torch.manual_seed(7 + rank)                      # ranks initialise differently on purpose
        m = _trainer(device, adv)
        net, optimizer, step, alpha = m._prepare(make_domain(1), make_domain(2, n=130, e=500))
        D.broadcast_parameters(net)                      # what _train_epochs does at the first epoch
        src = next(iter(m.source_loader)).to(device)
        tgt = next(iter(m.target_loader)).to(device)
        torch.manual_seed(100 + rank)                    # the MMD row draws of this rank
        loss, _, _ = m.forward_model(src, tgt, ALPHA)
        optimizer.zero_grad()
        loss.backward()
        D.allreduce_grads(p for g in optimizer.param_groups for p in g["params"])
        q.put((rank, dict(state={k: _np(v) for k, v in net.state_dict().items()},
                          grads={k: _np(p.grad) for k, p in net.named_parameters()},
                          src=_cpu(src), tgt=_cpu(tgt), loss=float(loss.detach()))))
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
    return [dict(state=_tensors(got[r]["state"]), grads=_tensors(got[r]["grads"]), src=_tensors(got[r]["src"]),
                 tgt=_tensors(got[r]["tgt"]), loss=got[r]["loss"]) for r in range(world)]


def concatenated_reference(results, device, adv, oracle):
    """The same objective on ONE process: union batch, concatenated row samples."""
    from pygda_amd.data import Data
    restore = inject_oracle() if oracle else None
    try:
        return _concatenated_reference(results, device, adv, Data)
    finally:
        if restore is not None:
            restore()


def _concatenated_reference(results, device, adv, Data):
    world = len(results)

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
    m.a2gnn = m.init_model().to(device)
    m.a2gnn.load_state_dict(results[0]["state"])
    m.a2gnn.train()
    m.a2gnn.zero_grad()
    if adv:
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
