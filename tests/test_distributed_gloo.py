"""CPU, world_size 2 over gloo: the data-parallel exchange steps of pygda_amd/distributed.py
(flat gradient all-reduce; all-gather of MMD sample rows with the slice-and-scale backward)
and the rank sharding of the loaders.  The kernels themselves need a GPU; the collective
logic is backend-independent and is what runs over RCCL on the GPU box."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pair_loss(rows_s, rows_t):
    """A differentiable stand-in for the global-batch pairwise statistic (same structure as
    MMD: every row interacts with every other row of both domains)."""
    tot = torch.cat([rows_s, rows_t], dim=1)                   # [times, 2n, d]
    d2 = ((tot.unsqueeze(1) - tot.unsqueeze(2)) ** 2).sum(-1)
    return torch.exp(-d2 / 4.0).mean()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pygda_amd import distributed as D
        from pygda_amd.data import Data, NeighborLoader
        assert D.active() and D.info() == dict(rank=rank, world_size=world)

        # ---- 1. flat gradient all-reduce == mean of the per-rank gradients -----------------
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
        unused = torch.nn.Parameter(torch.zeros(2))            # a parameter no rank touches
        g = torch.Generator().manual_seed(100 + rank)
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
        torch.nn.functional.cross_entropy(net(x), y).backward()
        local = [p.grad.clone() for p in net.parameters()]
        D.allreduce_grads(list(net.parameters()) + [unused])
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        for i, p in enumerate(net.parameters()):
            want = sum(gr[i] for gr in gathered) / world
            assert torch.allclose(p.grad, want, atol=1e-7)
        assert unused.grad is not None and torch.equal(unused.grad, torch.zeros(2))

        # ---- 1b. replicas start from rank 0's weights whatever each rank's seed was -----------
        torch.manual_seed(50 + rank)
        rep = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3))
        D.broadcast_parameters(rep)
        mine = torch.cat([t.reshape(-1) for t in rep.state_dict().values() if t.is_floating_point()])
        every = [None] * world
        dist.all_gather_object(every, mine)
        assert all(torch.equal(every[0], e) for e in every)

        # ---- 2. all-gather of sample rows: global loss, per-rank gradient slices ------------
        g = torch.Generator().manual_seed(7 + rank)
        rs = torch.randn(3, 5, 4, generator=g, requires_grad=True)      # [times, per, d] of this rank
        rt = torch.randn(3, 5, 4, generator=g, requires_grad=True)
        S = D.all_gather_rows(rs).permute(1, 0, 2, 3).reshape(3, world * 5, 4)
        Tt = D.all_gather_rows(rt).permute(1, 0, 2, 3).reshape(3, world * 5, 4)
        loss = _pair_loss(S, Tt)
        loss.backward()
        both = [None] * world
        dist.all_gather_object(both, (rs.detach(), rt.detach(), rs.grad, rt.grad, loss.item()))
        # single-process reference on the concatenated rows
        RS = torch.stack([b[0] for b in both]).permute(1, 0, 2, 3).reshape(3, world * 5, 4).requires_grad_()
        RT = torch.stack([b[1] for b in both]).permute(1, 0, 2, 3).reshape(3, world * 5, 4).requires_grad_()
        ref = _pair_loss(RS, RT)
        ref.backward()
        assert abs(ref.item() - loss.item()) < 1e-7 and all(abs(b[4] - loss.item()) < 1e-7 for b in both)
        want_s = RS.grad.view(3, world, 5, 4)[:, rank] * world         # scaled: allreduce_grads divides by W
        want_t = RT.grad.view(3, world, 5, 4)[:, rank] * world
        assert torch.allclose(rs.grad, want_s, atol=1e-7) and torch.allclose(rt.grad, want_t, atol=1e-7)

        # ---- 3. loaders: disjoint seed shards, equal step counts -----------------------------
        n = 100
        gg = torch.Generator().manual_seed(3)
        d = Data(x=torch.randn(n, 3, generator=gg), edge_index=torch.randint(0, n, (2, 400), generator=gg),
                 y=torch.zeros(n, dtype=torch.long))
        loader = NeighborLoader(d, [3, 3], batch_size=16, **D.info())
        seeds = [b.n_id[:b.batch_size].tolist() for b in loader]
        allseeds = [None] * world
        dist.all_gather_object(allseeds, seeds)
        assert len({len(s) for s in allseeds}) == 1                      # same number of steps
        flat = [tuple(s) for r in allseeds for s in r]
        assert set(v for s in flat for v in s) == set(range(n))          # union covers every seed
        q.put((rank, "ok"))
    except Exception as e:      # surface the failure in the parent
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.parametrize("adv", [False, True])
def test_two_rank_a2gnn_step_equals_concatenated_batch(adv):
    """SURVEY 8(e)'s equality test on the trainer itself: two ranks run ``A2GNN.forward_model`` on their
    own sampled mini-batches (sub-graphs of different sizes), exchange the MMD sample rows, weight their
    CE means by node counts and average their gradients; the result equals the gradient of ONE process
    on the concatenated batch.  CPU tensors, the oracle injected under the operator layer
    (tests/dp_equality.py) -- the same test runs on the HIP kernels in tests/test_gpu_configs.py."""
    from tests import dp_equality as E
    results = E.run_ranks(2, "cpu", adv, oracle=True)
    for k, v in results[0]["state"].items():                       # replicas started from rank 0's weights
        assert torch.equal(v, results[1]["state"][k]), k
    for k, v in results[0]["grads"].items():                       # one averaged gradient on every rank
        assert torch.equal(v, results[1]["grads"][k]), k
    ref_loss, ref_grads, (ns, nt) = E.concatenated_reference(results, "cpu", adv, oracle=True)
    assert ns[0] != ns[1] or nt[0] != nt[1]                        # the count weighting is exercised
    assert abs(results[0]["loss"] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    assert abs(results[1]["loss"] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    for k, g in ref_grads.items():
        assert torch.allclose(results[0]["grads"][k], g, rtol=1e-4, atol=1e-6), (k, (results[0]["grads"][k] - g).abs().max())


@pytest.mark.parametrize("kind", ["udagcn", "adagcn"])
def test_two_rank_udagcn_adagcn_step_equals_concatenated_batch(kind):
    """BASELINE.json configs[3] shards UDAGCN / AdaGCN mini-batches over ranks.  UDAGCN (udagcn.py:165-199): source CE,
    two gradient-reversed domain CEs and the entropy term are four means over sub-graphs of different sizes; AdaGCN
    (adagcn.py:169-198, 387-454): ten critic updates -- Wasserstein gap of two global means + gradient penalty as a
    mean over every rank's rows, critic gradients averaged before each Adam step -- then CE + |gap|.  Two ranks ==
    one process on the union batch: loss, every encoder gradient, and for AdaGCN the critic after its ten steps."""
    from tests import dp_equality as E
    results = E.run_ranks(2, "cpu", kind, oracle=True)
    for k, v in results[0]["state"].items():
        assert torch.equal(v, results[1]["state"][k]), k
    assert results[0]["grads"] and set(results[0]["grads"]) == set(results[1]["grads"])
    for k, v in results[0]["grads"].items():
        assert torch.equal(v, results[1]["grads"][k]), k
    ref_loss, ref_grads, (ns, nt, extra) = E.concatenated_reference(results, "cpu", kind, oracle=True)
    assert ns[0] != ns[1] or nt[0] != nt[1]                        # the count weighting is exercised
    for r in results:
        assert abs(r["loss"] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (r["loss"], ref_loss)
    assert set(ref_grads) == set(results[0]["grads"])
    for k, g in ref_grads.items():
        assert torch.allclose(results[0]["grads"][k], g, rtol=1e-4, atol=1e-6), (k, (results[0]["grads"][k] - g).abs().max())
    if kind == "adagcn":
        for k, v in extra["disc10"].items():                       # replica critics identical, and equal to the union's
            assert torch.equal(results[0]["disc10"][k], results[1]["disc10"][k]), k
            assert not torch.equal(results[0]["disc10"][k], results[0]["disc0"][k]), k      # it did train
            assert torch.allclose(results[0]["disc10"][k], v, rtol=1e-4, atol=1e-5), (k, (results[0]["disc10"][k] - v).abs().max())


# ---- bench.py's launcher: `--gpus N` must mean N ranks, or an error ------------------------------------------
def _bench(*argv, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                             "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=root)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the CPU form of the launcher test (gloo)")
def test_bench_spawns_the_ranks_it_is_asked_for():
    """VERDICT round 2, missing item 4: `python bench.py --gpus 2` without a launcher's environment used to run ONE
    rank and print n_gpus: 1.  It now starts the ranks itself (same torch.distributed.run command line the driver
    uses) and checks the group's size with an all-reduce; --launch-check stops there (gloo on a box without GPUs)."""
    import json
    r = _bench("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out == {"launch_check": "ok", "n_gpus": 2, "backend": "gloo"}


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box with fewer GPUs than asked for")
def test_bench_refuses_to_run_fewer_ranks_than_requested():
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout                       # no JSON line claiming a result
    assert "refusing" in r.stderr or "needs an MI355X" in r.stderr
    # a launcher whose world size disagrees with --gpus is an error as well
    r = _bench("--gpus", "4", "--launch-check", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_sampler_threads_divide_by_the_local_world_size(monkeypatch):
    from pygda_amd import sampler
    monkeypatch.delenv("PYGDA_AMD_SAMPLER_THREADS", raising=False)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)), raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert sampler.default_threads() == 16
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert sampler.default_threads() == 8                  # 8 ranks x 2 loaders x 8 = the host's 128 threads
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    assert sampler.default_threads() == 1
    monkeypatch.setenv("PYGDA_AMD_SAMPLER_THREADS", "3")
    assert sampler.default_threads() == 3


def test_direct_communicator_falls_back_when_it_cannot_be_built(monkeypatch):
    """distributed.direct_agreed(): the library-owned RCCL communicator is OPT-IN and built in stages, each followed by
    an agreement of all ranks; ANY failure (no librccl symbols, init error, failed self-test, a PEER's failure) is
    reported once, destroys what was built and leaves every collective on the torch.distributed ProcessGroup -- no
    second attempt, no exception into the training loop, and the same number of ProcessGroup collectives on a failing
    rank as on a healthy one (a rank failing alone must not desynchronise the group)."""
    import warnings
    from pygda_amd import distributed as D
    log = []

    class Comm:
        def __init__(self):
            log.append("new")
            self.handle = None

        def load(self):
            log.append("load")
            if fail_at == "load":
                raise RuntimeError("librccl symbols not found")

        def init_rank(self):
            log.append("init_rank")
            self.handle = 1
            if fail_at == "init_rank":
                raise RuntimeError("ncclCommInitRank failed")

        def destroy(self):
            log.append("destroy")

    def self_test(comm):
        log.append("self_test")
        if fail_at == "self_test":
            raise TimeoutError("no completion")

    agreements = []

    def all_ok(local_ok):
        agreements.append(local_ok)
        return local_ok and not peer_fails_at == len(agreements)

    monkeypatch.setattr(D, "active", lambda: True)
    monkeypatch.setattr(D.dist, "get_backend", lambda *a, **k: "nccl")
    monkeypatch.setattr(D, "_DirectComm", Comm)
    monkeypatch.setattr(D, "_self_test", self_test)
    monkeypatch.setattr(D, "_all_ranks_ok", all_ok)

    def fresh():
        monkeypatch.setattr(D, "_direct", None)
        monkeypatch.setattr(D, "_direct_failed", None)
        monkeypatch.setattr(D, "_direct_tried", False)
        log.clear()
        agreements.clear()

    # off unless asked for: never even attempted, and direct() never builds anything on its own
    monkeypatch.delenv("PYGDA_AMD_RCCL_DIRECT", raising=False)
    fail_at, peer_fails_at = None, 0
    fresh()
    assert D.direct_agreed() is None and D.direct() is None and log == [] and agreements == []
    monkeypatch.setenv("PYGDA_AMD_RCCL_DIRECT", "1")
    assert D.direct() is None and log == []
    # every way of failing on THIS rank: three agreements all the same (the peers' count), later stages skipped
    for fail_at, ran in (("load", ["new", "load", "destroy"]), ("init_rank", ["new", "load", "init_rank", "destroy"]),
                         ("self_test", ["new", "load", "init_rank", "self_test", "destroy"])):
        fresh()
        with pytest.warns(UserWarning, match="library-owned RCCL communicator unavailable"):
            assert D.direct_agreed() is None
        assert log == ran and fail_at in D._direct_failed
        assert len(agreements) == {"load": 1, "init_rank": 2, "self_test": 3}[fail_at] and agreements[-1] is False
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            assert D.direct_agreed() is None and D.direct() is None and D.capture_collectives() is False   # remembered
        assert log == ran
    # a PEER fails at stage 2 while this rank is healthy: this rank leaves at the same point and destroys its handle
    fail_at, peer_fails_at = None, 2
    fresh()
    with pytest.warns(UserWarning, match="another rank failed"):
        assert D.direct_agreed() is None
    assert log == ["new", "load", "init_rank", "destroy"] and agreements == [True, True]
    # all good
    peer_fails_at = 0
    fresh()
    comm = D.direct_agreed()
    assert comm is not None and D.direct() is comm and agreements == [True, True, True]
    assert log == ["new", "load", "init_rank", "self_test"]
