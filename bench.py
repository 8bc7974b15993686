#!/usr/bin/env python
"""Headline benchmark: A2GNN training-step throughput on ACMv9->DBLPv7 shapes.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the reference's training loop (pygda/models/a2gnn.py:308-319
plus the per-epoch metric :328-329) -- with full-batch loading that is one epoch.  Work unit
(SURVEY.md §8d): edges aggregated = sum over every executed aggregation (forward and
backward) of nnz(A_hat) incl. self loops = nnz_s*(4*L*s_pnums+2) + nnz_t*(3*L*t_pnums+1).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
``roofline`` (dominant kernel family, timed live with HIP events on the launch stream) and
``cpu_baseline`` (the CPU oracle's training step on the same workload, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
FP32_MFMA_PEAK_TF = 157.3      # v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TF = 2500.0      # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md: ~2.5 PF, measured 2178-2382)
LDS_READ_B32_PEAK_GBS = 128 * 256 * 2.4   # ds_read_b32: 128 B/clk/CU (MI355X_MICROARCH.md, LDS table) x 256 CUs x 2.4 GHz


def make_cfg_a(seed=200, ns=9360, es=15556, nt=5484, et=8117, feat=6775, classes=5, density=0.01,
               degrees="uniform", zipf=0.8):
    """Shape-identical stand-in for CitationDataset ACMv9 -> DBLPv7 (the files are not in the
    reference checkout, data/README.md links Google Drive only): E distinct undirected pairs
    symmetrised, x ~ Bernoulli(0.01) float32 [N, 6775], y uniform in {0..4}; seed 200 is the
    benchmark scripts' (unused) default seed (benchmark/node/a2gnn.py:23).
    ``degrees='powerlaw'``: same N and E, endpoints drawn with probability ~ rank^-zipf (a Chung-Lu graph on Zipf
    weights): a few hubs with several hundred neighbours over a majority of degree-1/2 nodes, which is what real
    citation graphs look like and what uniform pairs (maximum degree ~12) do not exercise."""
    from pygda_amd.data import Data
    g = torch.Generator().manual_seed(seed)

    def graph(n, e):
        keys = torch.empty(0, dtype=torch.int64)
        if degrees == "powerlaw":
            wgt = torch.arange(1, n + 1, dtype=torch.float64).pow(-zipf)[torch.randperm(n, generator=g)]
        while keys.numel() < e:
            if degrees == "powerlaw":
                a = torch.multinomial(wgt, 2 * e, replacement=True, generator=g)
                b = torch.multinomial(wgt, 2 * e, replacement=True, generator=g)
            else:
                a = torch.randint(0, n, (2 * e,), generator=g)
                b = torch.randint(0, n, (2 * e,), generator=g)
            lo, hi = torch.minimum(a, b), torch.maximum(a, b)
            k = (lo * n + hi)[lo != hi]
            keys = torch.unique(torch.cat([keys, k]))
        keys = keys[torch.randperm(keys.numel(), generator=g)[:e]]
        lo, hi = keys // n, keys % n
        ei = torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])
        x = (torch.rand(n, feat, generator=g) < density).float()
        y = torch.randint(0, classes, (n,), generator=g)
        return Data(x=x, edge_index=ei, y=y)

    return graph(ns, es), graph(nt, et)


def library_sha16():
    """First 16 hex digits of the SHA-256 of the loaded libgda_hip.so: what a committed rocprofv3 summary must carry
    (tools/summarize_rocprof.py records it) for its kernel durations to be quoted beside this run's without a warning."""
    import hashlib
    try:
        from pygda_amd._build import LIB
        return hashlib.sha256(open(LIB, "rb").read()).hexdigest()[:16]
    except Exception:                     # noqa: BLE001
        return None


def summary_library(fname):
    """The library hash a committed summary was made with (None: an older summary that does not record one)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", fname))).get("library_sha16")
    except Exception:                     # noqa: BLE001
        return None


def pmc_traffic(prefix="k_spmm<32, 4", pattern="*_rocprof_summary.json"):
    """HBM-side bytes per launch of the aggregation kernel from the committed rocprofv3 PMC passes
    (profiles/*_rocprof_summary.json, made by tools/summarize_rocprof.py: separate --pmc FETCH_SIZE
    and --pmc WRITE_SIZE runs of this same command, 2 x FETCH + WRITE per the gfx950 correction).
    PMC counters cannot be collected from inside the timed run; None if no summary is present."""
    import glob
    import re

    def rnd(f):          # newest ROUND first (file times do not survive a checkout)
        m = re.match(r"r(\d+)", os.path.basename(f))
        return (int(m.group(1)) if m else -1, os.path.basename(f))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=rnd)
    for f in reversed(files):
        try:
            pmc = json.load(open(f)).get("pmc", {})
        except Exception:
            continue
        for k, v in pmc.items():
            if k.startswith(prefix):
                return v["traffic_bytes"], os.path.basename(f)
    return None, None


def traffic_provenance(fname):
    """``{file, library_sha16, same_library_as_this_run}`` of a committed PMC summary: a counter
    figure printed beside this run's durations says which build of the library it was taken with."""
    if not fname:
        return None
    lib = summary_library(fname)
    return {"file": fname, "library_sha16": lib, "same_library_as_this_run": lib is not None and lib == library_sha16()}


def strict_fp32_companion(workload, steps, warmup, extra=()):
    """ms/step of the SAME workload with every product in strict fp32 arithmetic -- PYGDA_AMD_MMD_ONE_PASS=0 (the two-pass
    fp32-MFMA MMD kernels instead of the split-fp16 one-pass kernel) and PYGDA_AMD_GEMM_SPLIT_F16=0 (the fp32-MFMA tall
    products) -- measured in a child process (both switches are read once per process), short form of this script
    (VERDICT round 5, item 4b).  None if the child fails: a side figure must not cost the line."""
    import subprocess
    env = dict(os.environ, PYGDA_AMD_MMD_ONE_PASS="0", PYGDA_AMD_GEMM_SPLIT_F16="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-hbm-probe", "--no-side-lines", "--no-sustained", "--profile-run", "--no-strict-fp32",
           *extra]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "what": "same workload, same process layout, PYGDA_AMD_MMD_ONE_PASS=0 + PYGDA_AMD_GEMM_SPLIT_F16=0: every "
                        "product on fp32 MFMAs (the reference's own arithmetic type), child process"}
    except Exception as exc:              # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"[:200]}


def rocprof_kernel(prefix, pattern="*_rocprof_summary.json"):
    """``(avg_us, calls, file)`` of the first kernel whose name starts with ``prefix`` in the newest committed
    ``rocprofv3 --kernel-trace --stats`` summary matching ``pattern`` (profiles/, made by tools/profile_r4.sh from this
    same bench command): the in-graph average duration the driver's judge reads.  ``(None, None, None)`` if absent."""
    import glob
    import re

    def rnd(f):
        m = re.match(r"r(\d+)", os.path.basename(f))
        return (int(m.group(1)) if m else -1, os.path.basename(f))
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=rnd)):
        try:
            kernels = json.load(open(f)).get("kernels", [])
        except Exception:
            continue
        for k in kernels:
            if k["name"].startswith(prefix):
                return k["avg_us"], k["calls"], os.path.basename(f)
    return None, None, None


def max_row(g):
    """Longest row of the normalised adjacency (self loop included)."""
    rp = g.rowptr[:g.num_nodes + 1]
    return int((rp[1:] - rp[:-1]).max()) if g.num_nodes else 0


def edges_per_step(nnz_s, nnz_t, L, s_p, t_p):
    return nnz_s * (4 * L * s_p + 2) + nnz_t * (3 * L * t_p + 1)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(src, tgt, hp, edges, what="cfg-A"):
    """The CPU oracle (oracle/pygda_cpu.py, kind='port') on the same workload: full training steps (forward
    + backward + Adam), the reference's own op sequence incl. the [2000,2000,128] MMD temporaries.  One
    untimed warm-up step (the first call pays the allocator's page faults on those 2 GB temporaries: SURVEY
    6 measured ~10x), then ONE timed step: ~10-15 s each on the GPU box's host cores."""
    import psutil
    from oracle import pygda_cpu as O
    threads = torch.get_num_threads()
    torch.manual_seed(1)
    net = O.A2GNNBase(src.x.size(1), hp["hid"], hp["classes"], num_layers=hp["L"], dropout=hp["dropout"])
    opt = torch.optim.Adam(net.parameters(), lr=hp["lr"], weight_decay=hp["wd"])
    s, t = O.Graph(src.x, src.edge_index, src.y), O.Graph(tgt.x, tgt.edge_index, tgt.y)
    chunk = None if psutil.virtual_memory().available > 48 * 2 ** 30 else 128
    t0 = time.perf_counter()
    O.a2gnn_train_step(net, opt, s, t, 0.0, hp["s_pnums"], hp["t_pnums"], False, hp["weight"], chunk)
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.a2gnn_train_step(net, opt, s, t, 0.0, hp["s_pnums"], hp["t_pnums"], False, hp["weight"], chunk)
    dt = time.perf_counter() - t0
    return {"value": edges / dt, "unit": "edges/s", "cores": threads, "cpu": cpu_model(), "kind": "port",
            "edges_per_step": edges,
            "sample": f"1 timed A2GNN training step (fwd+bwd+Adam) of the same {what} workload after 1 untimed "
                      f"warm-up step ({warm:.1f} s): {dt:.1f} s" + ("" if chunk is None else ", MMD temporaries row-chunked"),
            "steps_per_sec": 1.0 / dt}


def hbm_regime_probe(device, nodes=5_000_000, avg_degree=20, d=128, iters=5):
    """The aggregation kernel where its roofline is HBM: ONE full-graph SpMM at BASELINE.json configs[4]'s
    per-domain size (5 M nodes, 100 M directed edges + self loops, d = 128: the feature matrix alone is 2.56 GB,
    ten times the Infinity Cache), timed with HIP events in this process.  `achieved` uses SURVEY 8(d)'s
    algorithmic bytes nnz*8 + (N+1)*4 + 2*N*d*4; `gather_model_GBs` is the no-reuse figure nnz*(8+4d) + N*d*4
    (uniform random neighbours share nothing, so every neighbour row is a distinct 512-byte read): the rate
    the memory system actually sustains.  Counter traffic comes from the committed PMC passes of
    tools/spmm_sweep.py --big (profiles/r2_spmm5m_rocprof_summary.json)."""
    from pygda_amd import ops
    from pygda_amd.graph import build_csr
    gen = torch.Generator(device=device).manual_seed(200)
    half = nodes * avg_degree // 2
    a = torch.randint(0, nodes, (half,), generator=gen, device=device)
    b = torch.randint(0, nodes, (half,), generator=gen, device=device)
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    del a, b
    G = build_csr(ei, nodes, validate=False)
    nnz = G.nnz
    x = torch.randn(nodes, d, device=device, generator=gen)
    for _ in range(2):
        y = ops.spmm_kstep(G, x, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        y = ops.spmm_kstep(G, x, 1)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    alg = nnz * 8 + (nodes + 1) * 4 + 2 * nodes * d * 4
    gather = nnz * (8 + 4 * d) + nodes * d * 4
    traffic, src_file = pmc_traffic("k_spmm<32, 4", "r[0-9]*_hbm_uniform*_summary.json")      # tools/profile.sh <round> hbm
    if traffic is None:
        traffic, src_file = pmc_traffic("k_spmm<32, 4", "r2_spmm5m*_summary.json")
    del G, x, y, ei
    torch.cuda.empty_cache()
    out = {"kernel": f"spmm_csr_f32[d={d}] (k_spmm<32,4>), N={nodes}, nnz={nnz}", "bound": "hbm",
           "achieved": alg / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / us / 1e3 / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": src_file, "traffic_provenance": traffic_provenance(src_file),
           "avg_launch_us": us, "launches": iters,
           "algorithmic_bytes_per_launch": alg, "gather_model_bytes_per_launch": gather,
           "gather_model_GBs": gather / us / 1e3, "gather_model_frac": gather / us / 1e3 / HBM_PEAK_GBS,
           "traffic_over_algorithmic": (traffic / alg) if traffic else None,
           "what_the_waste_is": "uniform random neighbours share nothing: every one of the nnz neighbour rows is a distinct "
                                "512-byte read (the gather model), ten Infinity Caches of feature rows -- counter traffic "
                                "equals the gather model to 0.3 %, i.e. the kernel runs at the memory system's limit on "
                                "9.6 x the algorithmic bytes",
           "timing": "HIP events on the launch stream, same process, after the timed region"}
    out["rmat_2^22"] = rmat_probe(device, d, iters)
    return out


def rmat_probe(device, d=128, iters=5):
    """The same full-graph aggregation on a power-law graph (R-MAT a, b, c = 0.57, 0.19, 0.19; 2^22 nodes, 32 M edges
    symmetrised), as generated and after a degree-sorted relabelling (hubs first: the rows most rows gather share
    cache lines and pages) -- the locality option a uniform graph has no use for (profiles/HISTORY.md 5, round 3)."""
    from pygda_amd import ops
    from pygda_amd.graph import build_csr
    from tools.spmm_sweep import rmat_edges
    gen = torch.Generator(device=device).manual_seed(200)
    n = 1 << 22
    ei = rmat_edges(22, 32_000_000, gen)
    ei = torch.cat([ei, ei.flip(0)], dim=1)
    res = {}
    from pygda_amd.data import Data, auto_reorder
    for name in ("as_generated", "trainer_default"):
        if name == "trainer_default":
            # what A2GNN.fit() trains on BY DEFAULT for a full-batch graph of this size and skew: the loader applies
            # data.auto_reorder (degree-ordered relabelling, predict() maps the rows back) -- no opt-in by the caller
            probe = Data(x=torch.empty(n, 1, device=device), edge_index=ei, y=None)
            relabelled, new_id = auto_reorder(probe)
            res["trainer_default_reorders"] = new_id is not None
            ei = relabelled.edge_index
            del probe, relabelled, new_id
        G = build_csr(ei, n, validate=False)
        x = torch.randn(n, d, device=device, generator=gen)
        for _ in range(2):
            ops.spmm_kstep(G, x, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            ops.spmm_kstep(G, x, 1)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / iters
        alg = G.nnz * 8 + (n + 1) * 4 + 2 * n * d * 4
        gather = G.nnz * (8 + 4 * d) + n * d * 4          # the no-reuse gather model: every neighbour row a distinct read
        res[name] = {"avg_launch_us": us, "nnz": G.nnz, "algorithmic_GBs": alg / us / 1e3, "frac": alg / us / 1e3 / HBM_PEAK_GBS,
                     "gather_model_over_algorithmic": gather / alg, "gather_model_GBs": gather / us / 1e3}
        # counter traffic of the same aggregation from the committed PMC passes (tools/rmat_pmc_case.py under separate
        # --pmc FETCH_SIZE / WRITE_SIZE runs): how much of the gather model the caches absorb on a skewed graph
        which = "asgen" if name == "as_generated" else "reorder"
        traffic, tfile = pmc_traffic("k_spmm<32, 4", "r[0-9]*_hbm_" + which + "*_summary.json")
        if traffic is None:
            traffic, tfile = pmc_traffic("k_spmm<32, 4", "r[0-9]*_rmat_" + which + "*_summary.json")
        if traffic:
            res[name].update(traffic_bytes_per_launch=traffic, traffic_source=tfile, traffic_over_algorithmic=traffic / alg,
                             traffic_provenance=traffic_provenance(tfile))
        del G, x
        torch.cuda.empty_cache()
    return res


def make_cfg_s(nodes, avg_degree, feat, classes, seed, device):
    """One domain of BASELINE.json configs[4]: `nodes` nodes, nodes*avg_degree directed edges
    (uniform random pairs, both directions), x ~ N(0,1) fp32 [nodes, feat] resident on the GPU."""
    from pygda_amd.data import Data
    g = torch.Generator(device=device).manual_seed(seed)
    half = nodes * avg_degree // 2
    a = torch.randint(0, nodes, (half,), generator=g, device=device)
    b = torch.randint(0, nodes, (half,), generator=g, device=device)
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    x = torch.randn(nodes, feat, generator=g, device=device)
    y = torch.randint(0, classes, (nodes,), generator=g, device=device)
    return Data(x=x, edge_index=ei, y=y)


def run_cfg_s(args, world, rank, dev, cpu_base=True):
    """Sampled mini-batch A2GNN on the synthetic large graphs (configs[4]): the neighbour sampler
    (prefetching), the row-gather kernel, per-batch graph ingestion, and -- with N ranks -- disjoint
    seed shards, all-gathered MMD rows and one flat gradient all-reduce per step.  Returns the JSON
    object on rank 0 (None elsewhere)."""
    import torch.distributed as dist
    from pygda_amd import ops
    from pygda_amd.models import A2GNN
    hp = dict(hid=128, classes=5, L=2, lr=0.01, wd=0.005, dropout=0.5, s_pnums=0, t_pnums=10, weight=10)
    fan = [int(v) for v in args.fanout.split(",")]
    src = make_cfg_s(args.nodes, args.avg_degree, args.feat, hp["classes"], 200, dev)
    tgt = make_cfg_s(args.nodes, args.avg_degree, args.feat, hp["classes"], 201, dev)
    steps_total = args.warmup + args.steps
    # one epoch = steps_total global steps: every rank needs that many batches of `batch` seeds
    need = steps_total * args.batch * world
    gsel = torch.Generator().manual_seed(7)
    seeds_s = torch.randint(0, args.nodes, (need,), generator=gsel)
    seeds_t = torch.randint(0, args.nodes, (need,), generator=gsel)
    from pygda_amd.data import NeighborLoader
    model = A2GNN(args.feat, hp["hid"], hp["classes"], num_layers=hp["L"], lr=hp["lr"], weight_decay=hp["wd"],
                  epoch=1, dropout=hp["dropout"], s_pnums=hp["s_pnums"], t_pnums=hp["t_pnums"],
                  weight=hp["weight"], device=dev, batch_size=args.batch, num_neigh=fan, verbose=0)
    torch.manual_seed(1234 + rank)
    net, optimizer, step_fn, alpha_fn = model._prepare(src, tgt)
    if world > 1:                # replicas of ONE model (fit() does this in its epoch loop; the generators stay per rank)
        from pygda_amd.distributed import broadcast_parameters
        broadcast_parameters(net)
    kw = dict(rank=rank, world_size=world, device=dev, recycle=True)         # as BaseGDA._node_loaders builds them
    model.source_loader = NeighborLoader(src, fan, batch_size=args.batch, input_nodes=seeds_s, **kw)
    model.target_loader = NeighborLoader(tgt, fan, batch_size=args.batch, input_nodes=seeds_t, **kw)
    if args.eager:
        model.use_hip_graph = False
    # the step replayed at ONE static shape (pygda_amd/sampled_graph.py: what fit() does for sampled A2GNN training on
    # one GPU); None -> eager launches (--eager, data-parallel runs, PYGDA_AMD_SAMPLED_GRAPH=0)
    model._declare_static_shape(model.source_loader, model.target_loader)
    stepper = model._sampled_stepper(net, optimizer, step_fn)
    raw_its = (model.source_loader.iter_raw(), model.target_loader.iter_raw()) if stepper is not None else (None, None)
    if raw_its[0] is None or raw_its[1] is None:
        stepper = None
    it = zip(*raw_its) if stepper is not None else zip(iter(model.source_loader), iter(model.target_loader))
    from pygda_amd.models.base import _allreduce_grads
    step_sizes = []          # captured steps log no aggregation calls: their edge counts come from the batches' live sizes

    phases = []              # per step: host seconds in (loader hand-over, forward, backward, exchange + optimiser)

    stall_ms = float(os.environ.get("PYGDA_AMD_BENCH_STALL_TRACE", "0"))     # diagnosis: dump every thread's stack when a step
    if stall_ms > 0:                                                           # takes longer than this on the host
        import faulthandler

    def one_step():
        if stall_ms > 0:
            faulthandler.dump_traceback_later(stall_ms * 1e-3, exit=False)     # a C watchdog thread: needs no interpreter lock
        try:
            return one_step_()
        finally:
            if stall_ms > 0:
                faulthandler.cancel_dump_traceback_later()

    def one_step_():
        h0 = time.perf_counter()
        s, t = next(it)
        h1 = time.perf_counter()
        if stepper is not None:
            (ps, zs), (pt, zt) = s, t
            ticket = stepper.step(ps, zs, pt, zt)
            if ticket is not None:
                step_sizes.append((zs, ps.T_int, zt, pt.T_int))
                phases.append((h1 - h0, time.perf_counter() - h1, 0.0, 0.0))
                return stepper.stats
            s = model.source_loader._sampler.assemble(model.source_loader.data, ps, zs)      # a pair the static shape
            t = model.target_loader._sampler.assemble(model.target_loader.data, pt, zt)      # cannot take: eager, real shape
        ops.dropout_state.next_step(s.x.device)
        net.train()
        loss, _ = step_fn(s, t, 0.0, 0)
        h2 = time.perf_counter()
        optimizer.zero_grad()
        loss.backward()
        h3 = time.perf_counter()
        _allreduce_grads(optimizer)
        optimizer.step()
        if stepper is not None:          # an eager fall-back of the captured loop: hand the raw ring blocks back
            model.source_loader._sampler.release(ps)
            model.target_loader._sampler.release(pt)
        phases.append((h1 - h0, h2 - h1, h3 - h2, time.perf_counter() - h3))
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    if os.environ.get("PYGDA_AMD_SWITCH_US"):     # experiment: the interpreter's thread switch interval (default 5 ms)
        sys.setswitchinterval(1e-6 * float(os.environ["PYGDA_AMD_SWITCH_US"]))
    gc.collect()                 # nothing of an earlier workload in this process (captured graphs) dies inside the region
    for _ in range(args.warmup):
        one_step()
    ops.aggregation_log = []
    sync()
    phases.clear()
    step_sizes.clear()
    if stepper is not None:
        stepper.host_wait_s = stepper.host_work_s = 0.0
    gc.disable()                 # a generation-2 pass of the cyclic collector is milliseconds; the steps are 3 ms
    ms0 = torch.cuda.memory_stats(dev)
    allocs0 = ms0.get("num_device_alloc", 0)
    t0 = time.perf_counter()
    c0 = time.thread_time()
    marks, cpu_marks = [], []
    cprof = None
    if os.environ.get("PYGDA_AMD_BENCH_CPROFILE"):     # experiment: where the training thread's host time goes
        import cProfile
        cprof = cProfile.Profile()
        cprof.enable()
    for _ in range(args.steps):
        loss = one_step()
        marks.append(time.perf_counter())
        cpu_marks.append(time.thread_time())
    if cprof is not None:
        cprof.disable()
        import pstats
        with open(os.environ["PYGDA_AMD_BENCH_CPROFILE"], "w") as fh:
            st = pstats.Stats(cprof, stream=fh)
            st.sort_stats("tottime").print_stats(60)
            st.sort_stats("cumulative").print_stats(70)
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    host_ms = [1e3 * (b - a) for a, b in zip([t0] + marks[:-1], marks)]        # host time per step (enqueue side)
    # ... and how much of it the training thread spent ON a core: the rest is waiting (the interpreter lock held by the
    # loaders' producer threads / the MMD helper, a full launch queue, the loader's queue)
    host_cpu_ms = [1e3 * (b - a) for a, b in zip([c0] + cpu_marks[:-1], cpu_marks)]
    timed_phases = phases[:args.steps]
    worst = max(range(len(timed_phases)), key=lambda i: sum(timed_phases[i])) if timed_phases else None
    med = lambda k: sorted(p[k] for p in timed_phases)[len(timed_phases) // 2] * 1e3
    host_phases = None if worst is None else {
        "median_ms": {"loader": med(0), "forward": med(1), "backward": med(2), "exchange_optimiser": med(3)},
        "slowest_step": {"index": worst, "loader": 1e3 * timed_phases[worst][0], "forward": 1e3 * timed_phases[worst][1],
                         "backward": 1e3 * timed_phases[worst][2], "exchange_optimiser": 1e3 * timed_phases[worst][3]}}
    ms1 = torch.cuda.memory_stats(dev)
    device_allocs = ms1.get("num_device_alloc", 0) - allocs0   # hipMalloc calls inside the region
    if os.environ.get("PYGDA_AMD_BENCH_ALLOC_DEBUG") == "1":
        sys.stderr.write("alloc debug: " + json.dumps({k: ms1.get(k, 0) - ms0.get(k, 0) for k in (
            "num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.current",
            "allocated_bytes.all.peak", "allocation.all.allocated", "segment.all.allocated", "num_sync_all_streams")})
            + f" reserved now {ms1.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB\n")
    log, ops.aggregation_log = ops.aggregation_log, None
    # `edges_ref`: SURVEY 8(d)'s reference-equivalent count (every call = K full aggregations of the batch's nnz);
    # `edges`: the entries whose multiply-add the step really executed (the interior-rows paths never redo the leaf rows'
    # unit self loops nor the leaf columns' constant contribution per step) -- `value` is the executed one
    edges_ref = sum(e[0].nnz * e[1] for e in log)
    edges = sum((e[2] if len(e) > 2 and e[2] is not None else e[0].nnz * e[1]) for e in log)
    for zs, ts_, zt, tt_ in step_sizes:          # the replayed steps, from their batches' live sizes
        r, d_ = stepper.edges_of(zs, ts_, zt, tt_)
        edges_ref += r
        edges += d_
    if args.profile_run:
        if rank != 0:
            return None
        return {"metric": "edges_aggregated_per_sec", "value": edges / dt, "unit": "edges/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "profile_run": True}
    # the HIP-event pass below brackets kernel families on their launch stream: with the two branches on two streams an
    # event pair also spans the other stream's kernels, so this pass runs the step on ONE stream
    timed_loaders = (model.source_loader, model.target_loader)
    model.overlap_sampled = False
    # roofline inputs: a short pass of the same steps with HIP events around every kernel family (the timed
    # region above runs without them)
    from pygda_amd import profiler
    prof_steps = max(1, min(5, args.steps))
    extra_seeds = prof_steps * args.batch * world
    model.source_loader = NeighborLoader(src, fan, batch_size=args.batch,
                                         input_nodes=torch.randint(0, args.nodes, (extra_seeds,), generator=gsel), **kw)
    model.target_loader = NeighborLoader(tgt, fan, batch_size=args.batch,
                                         input_nodes=torch.randint(0, args.nodes, (extra_seeds,), generator=gsel), **kw)
    it = zip(iter(model.source_loader), iter(model.target_loader))
    last = {}

    def one_step_keep():
        s, t = next(it)
        last["s"], last["t"] = s, t
        ops.dropout_state.next_step(s.x.device)
        net.train()
        loss, _ = step_fn(s, t, 0.0, 0)
        optimizer.zero_grad()
        loss.backward()
        _allreduce_grads(optimizer)
        optimizer.step()

    one_step = one_step_keep
    sync()
    profiler.start()
    for _ in range(prof_steps):
        one_step()
    sync()
    profiler.stop()
    prof = profiler.summary()
    if world > 1:
        t = torch.tensor([dt, float(edges), float(edges_ref)], device=dev, dtype=torch.float64)
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, edges, edges_ref = float(tm[0]), float(t[1]), float(t[2])
    if rank == 0:
        def roof(name):
            r = prof[name]
            secs = r["ms"] * 1e-3
            if name.startswith("spmm") or name.startswith("kstep_lds") or name.startswith("interior_lds"):
                ach = r["bytes"] / secs / 1e9
                kern = ("k_spmm_range<32, 4" if name.startswith("spmm_interior") else
                        "k_il_lds" if name.startswith("interior_lds") else "k_spmm<32, 4")
                traffic, src_file = pmc_traffic(kern, "r[0-9]*_cfgS*_summary.json") if "d=128" in name else (None, None)
                out = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src_file,
                       "launches": r["launches"], "avg_launch_us": r["avg_us"],
                       "algorithmic_bytes_per_launch": r["bytes"] / r["launches"]}
                if name.startswith("spmm_interior") and "d=128" in name:
                    # the committed rocprofv3 averages of the two kernels behind this region price the same launches
                    ru, _, rf = rocprof_kernel("k_spmm_range<32, 4", "r[0-9]*_cfgS*_summary.json")
                    cu, _, _ = rocprof_kernel("k_rows_copy_bias", "r[0-9]*_cfgS*_summary.json")
                    if ru and cu:
                        copies = r.get("copy_launches", 0)
                        rp_secs = (ru * (r["launches"] - copies) + cu * copies) * 1e-6
                        out["rocprof"] = {"k_spmm_range_avg_us": ru, "k_rows_copy_bias_avg_us": cu, "source": rf,
                                          "what": "committed rocprofv3 averages: a labelled side figure, `frac` above is "
                                                  "this run's HIP-event duration",
                                          "same_library_as_this_run": summary_library(rf) is not None
                                          and summary_library(rf) == library_sha16(),
                                          "region_us_from_rocprof": rp_secs * 1e6, "region_us_live": secs * 1e6,
                                          "frac_rocprof": r["bytes"] / rp_secs / 1e9 / HBM_PEAK_GBS}
                if r.get("alg_equiv_bytes"):
                    out["bytes_are"] = ("what the interior-rows K-step itself has to move (K steps over the rows that can "
                                        "change, every distinct row read once per step, + one pass over the leaves), NOT "
                                        "K full aggregations; `launches` = K interior steps + the leaf pass")
                    out["k_full_aggregations_equivalent_GBs"] = r["alg_equiv_bytes"] / secs / 1e9
                    # SURVEY 8(d)'s definition side by side: K full aggregations' algorithmic bytes per call
                    out["survey_8d_bytes_per_call"] = r["alg_equiv_bytes"] / max(r["calls"], 1)
                    out["interior_rows_bytes_per_call"] = r["bytes"] / max(r["calls"], 1)
                    out["frac_survey_8d"] = r["alg_equiv_bytes"] / secs / 1e9 / HBM_PEAK_GBS
                    out["frac_is"] = ("interior-rows bytes (what this kernel must move) over its duration; frac_survey_8d "
                                      "prices the same duration with SURVEY 8(d)'s bytes of the K full aggregations the "
                                      "call stands for.  The batch (~80 MB of features) is Infinity-Cache resident: both "
                                      "are accounting figures, the kernel's counter traffic is in `traffic`")
                return out
            ach = r["flops"] / secs / 1e12
            split16 = (os.environ.get("PYGDA_AMD_GEMM_SPLIT_F16", "1") != "0"
                       and (name.startswith("dense_projection[") or name.startswith("dense_projection_dgrad["))
                       and all(v in ("128", "256") for v in name.split("[")[1].rstrip("]").split("x")))
            if split16:
                # k_tall_fwd_h (csrc/gda_gemm_split.inc): three fp16 MFMAs on split operands per fp32-equivalent product
                # -- priced against the roof of the instructions it EXECUTES; the fp32-equivalent figure stands beside it
                return {"kernel": name, "bound": "mfma", "executes": "v_mfma_f32_32x32x16_f16 on split operands (hi.hi + "
                        "hi.lo + lo.hi): 3 x the fp32-equivalent flops", "achieved": 3 * ach, "peak": F16_MFMA_PEAK_TF,
                        "unit": "TFLOP/s", "frac": 3 * ach / F16_MFMA_PEAK_TF, "traffic": None, "launches": r["launches"],
                        "avg_launch_us": r["avg_us"],
                        "fp32_equivalent": {"achieved": ach, "frac_of_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TF},
                        "note": "memory-bound at these shapes (a [rows, 128] operand in, one out): the MFMA fraction is small "
                                "by construction"}
            return {"kernel": name, "bound": "mfma", "executes": "v_mfma_f32_*_f32", "achieved": ach,
                    "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": ach / FP32_MFMA_PEAK_TF, "traffic": None, "launches": r["launches"],
                    "avg_launch_us": r["avg_us"]}
        agg = [k for k in prof if k.startswith("spmm") or k.startswith("kstep_lds") or k.startswith("interior_lds")]
        cands = agg + [k for k in prof if k.startswith("dense")]
        dominant = max(cands, key=lambda k: prof[k]["ms"])
        out = {
            "metric": "edges_aggregated_per_sec", "value": edges / dt, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": ("fp32 values everywhere; the MMD's two pair products run as three fp16 MFMAs on split operands (hi + lo, 22 "
                       "significant bits, fp32 accumulation) when the one-pass kernel covers the shape -- error against float64 "
                       "equal to the fp32-MFMA kernels' (profiles/r4_mmd_one_pass_experiments.txt); PYGDA_AMD_MMD_ONE_PASS=0 "
                       "runs those instead"),
           
            "config": {"workload": f"cfg-S: A2GNN on synthetic source+target graphs, {args.nodes} nodes / "
                                   f"{args.nodes * args.avg_degree} directed edges per domain, F={args.feat}, nhid=128, "
                                   f"L=2, s_pnums=0, t_pnums=10, NeighborLoader fan-out {fan}, {args.batch} seeds per GPU "
                                   "per step, MMD domain loss",
                       "edges_aggregated_per_step": edges / args.steps,
                       "edges_aggregated_per_step_reference_equivalent": edges_ref / args.steps,
                       "note": "value counts the entries whose multiply-add the step EXECUTES: the interior-rows "
                               "K-step paths do the leaf rows' unit self loops and the leaf columns' contribution once per "
                               "call, not once per step; reference_equivalent counts every call as K full aggregations",
                       "final_loss": float(loss.detach().reshape(-1)[0]),
                       "execution": ("eager launches" if stepper is None else
                                     "hipGraph replay of the step captured at its static shape ("
                                     + ("source branch + logits pass on side streams" if os.environ.get("PYGDA_AMD_SAMPLED_GRAPH_FORK", "1") == "1" else "single stream") + "): "
                                     f"{stepper.replays} replays, {stepper.fallbacks} eager fall-backs; rows per matrix "
                                     f"{stepper.static[0].ncap} + {stepper.static[1].ncap} (capacity) for "
                                     f"{int(sum(z[0][0] for z in step_sizes) / max(len(step_sizes), 1))} + "
                                     f"{int(sum(z[2][0] for z in step_sizes) / max(len(step_sizes), 1))} live ones"
                                     if stepper.static is not None else "eager launches (static shape declined)"),
                       "host_ms_per_step_max_median": [max(host_ms), sorted(host_ms)[len(host_ms) // 2]],
                       "host_cpu_ms_per_step_median": sorted(host_cpu_ms)[len(host_cpu_ms) // 2],
                       "host_phases": host_phases,
                       "producer_cpu_ms_per_batch": [1e3 * getattr(l, "producer_cpu_s", 0.0) / max(getattr(l, "producer_batches", 0), 1)
                                                     for l in timed_loaders],
                       "producer_enqueue_cpu_ms_per_batch": [1e3 * getattr(l, "producer_enqueue_cpu_s", 0.0) / max(getattr(l, "producer_batches", 0), 1)
                                                             for l in timed_loaders],
                       "hipMalloc_calls_in_timed_region": device_allocs,
                       # captured steps: the training thread's time inside step() split into event waits (the device is
                       # behind: back-pressure, not work) and everything else -- draws, counting sorts, two block copies,
                       # the graph launch.  `host_ms_per_step_max_median` above includes the waits.
                       "host_work_ms_per_step": None if stepper is None else 1e3 * stepper.host_work_s / max(len(step_sizes), 1),
                       "host_wait_ms_per_step": None if stepper is None else 1e3 * stepper.host_wait_s / max(len(step_sizes), 1),
                       "aggregation_launches_per_step": sum(prof[k]["launches"] for k in agg) / prof_steps,
                       "aggregation_paths": {k: prof[k]["launches"] / prof_steps for k in sorted(agg)},
                       "sampler": model.source_loader.sampler_description(),
                       "parallelism": "single GPU" if world == 1 else
                       f"dp{world}: disjoint seed mini-batches per rank, graph + features replicated, all-gathered "
                       "global-batch MMD rows, one flat RCCL gradient all-reduce per step"},
            "steps_per_sec": args.steps / dt,
            "reference_equivalent_edges_per_sec": edges_ref / dt,
            "roofline": dict(roof(dominant), timing=f"HIP events on the launch stream, {prof_steps} extra steps after "
                                                    "the timed region"),
            "roofline_dense_projection": {k: roof(k) for k in sorted(prof) if k.startswith("dense_projection")},
            "kernel_time_ms_per_step": {k: v["ms"] / prof_steps for k, v in sorted(prof.items())},
            "library_sha16": library_sha16()}
        if world == 1 and cpu_base and not args.no_cpu_baseline:
            # the oracle's training step on ONE sampled batch pair of this run (its sub-graphs, its features)
            sb, tb = last["s"], last["t"]
            from pygda_amd.data import Data
            cs = Data(x=sb.x.cpu(), edge_index=sb.edge_index.cpu(), y=sb.y.cpu())
            ct = Data(x=tb.x.cpu(), edge_index=tb.edge_index.cpu(), y=tb.y.cpu())
            from oracle import pygda_cpu as O
            nnz_s = O.gcn_norm(cs.edge_index, None, cs.x.size(0))[0].size(1)
            nnz_t = O.gcn_norm(ct.edge_index, None, ct.x.size(0))[0].size(1)
            out["cpu_baseline"] = cpu_baseline(cs, ct, hp, edges_per_step(nnz_s, nnz_t, hp["L"], hp["s_pnums"],
                                                                         hp["t_pnums"]),
                                               what=f"cfg-S sampled batch pair ({cs.x.size(0)} + {ct.x.size(0)} nodes)")
        return out
    return None


def relaunch(args):
    """``python bench.py --gpus N`` without a launcher's environment: start the N ranks here, through the same
    ``python -m torch.distributed.run`` command line the driver uses, and hand its exit code back.  A node with
    fewer than N GPUs is an error -- never a silent 1-rank run that prints ``n_gpus: 1``."""
    import socket
    import subprocess
    if (not args.launch_check or torch.cuda.is_available()) and not args.share_gpus:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible on this node; refusing "
                             "to run fewer ranks than requested")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def init_group(args, world, rank, dev):
    """RCCL (``nccl``) process group of exactly ``--gpus`` ranks; gloo only for ``--launch-check`` on a box
    without GPUs.  RCCL prints a version banner through C stdio on stdout when the communicator comes up; stdout
    is kept for the ONE JSON line, so fd 1 is routed to stderr until the banner has been flushed."""
    import ctypes
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ["PYGDA_AMD_FORCE_DP"] = "1"
    if args.rccl_direct:           # + the whole data-parallel step captured WITH its collectives (1-rank validated only)
        os.environ["PYGDA_AMD_RCCL_DIRECT"] = "1"
        os.environ["PYGDA_AMD_RCCL_CAPTURE"] = "1"
    if args.no_rccl_direct:
        os.environ["PYGDA_AMD_RCCL_DIRECT"] = "0"
    gpu = torch.cuda.is_available()
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        if gpu and not args.share_gpus:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
        warm = torch.ones(1, device=dev if gpu else "cpu")
        dist.all_reduce(warm)
        if gpu:
            torch.cuda.synchronize()
        if float(warm) != float(world):
            raise SystemExit(f"bench.py: all-reduce over the group summed to {float(warm)}, expected {world}")
        # who is there: an all-gather of the ranks (the line prints it as rccl_ranks_seen)
        mine = torch.tensor([rank], device=dev if gpu else "cpu", dtype=torch.int64)
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        args._ranks_seen = sorted(int(v) for v in torch.cat(seen).cpu())
        if args._ranks_seen != list(range(world)):
            raise SystemExit(f"bench.py: the group's ranks are {args._ranks_seen}, expected 0..{world - 1}")
        args._collectives = "torch.distributed ProcessGroup (" + dist.get_backend() + ")"
        if gpu and dist.get_backend() == "nccl":
            # the library-owned RCCL communicator is opt-in (--rccl-direct / PYGDA_AMD_RCCL_DIRECT=1: it has never seen
            # two ranks); when asked for it is kept only if EVERY rank built it and passed its self-test
            # (pygda_amd/distributed.py: staged, fault-symmetric), else all ranks stay on the ProcessGroup
            from pygda_amd import distributed as _D
            if _D.direct_agreed() is not None:
                args._collectives = "library-owned RCCL communicator (gda_comm_*: enqueues on the training stream)"
            elif _D._direct_failed:
                args._collectives += "; library-owned communicator declined: " + _D._direct_failed
            torch.cuda.synchronize()
    finally:
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved_fd, 1)
        os.close(saved_fd)


def run_cfg_a(args, world, rank, dev, side=False):
    """cfg-A (BASELINE.json configs[1]): full-batch A2GNN on the ACMv9 -> DBLPv7 shapes.  With ``side`` (the N > 1
    run, where cfg-A is one full-batch REPLICA per GPU) only a short labelled summary is returned."""
    import torch.distributed as dist
    from pygda_amd import profiler
    from pygda_amd.models import A2GNN
    # hyper-parameters of benchmark/node/run_citation.sh:92 (A2GNN, ACMv9 -> DBLPv7)
    hp = dict(hid=128, classes=5, L=2, lr=0.01, wd=0.005, dropout=0.5, s_pnums=0, t_pnums=10, weight=10)
    src, tgt = make_cfg_a(seed=200, degrees=args.graph)
    total_epochs = args.warmup + args.steps
    model = A2GNN(src.x.size(1), hp["hid"], hp["classes"], num_layers=hp["L"], lr=hp["lr"],
                  weight_decay=hp["wd"], epoch=total_epochs, dropout=hp["dropout"], s_pnums=hp["s_pnums"],
                  t_pnums=hp["t_pnums"], weight=hp["weight"], adv=args.adv, device=dev, verbose=0,
                  use_hip_graph=False if args.eager else None)   # None: captured, with the eager fallback if capture fails
    torch.manual_seed(1234 + rank)
    state = model._prepare(src, tgt)
    src_d, tgt_d = src.to(dev), tgt.to(dev)          # inputs resident in HBM before the timed region

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    model._train_epochs(*state, epochs=range(args.warmup))
    from pygda_amd.graph import as_graph
    nnz_s = as_graph(src_d.edge_index, src_d.num_nodes).nnz
    nnz_t = as_graph(tgt_d.edge_index, tgt_d.num_nodes).nnz
    edges = edges_per_step(nnz_s, nnz_t, hp["L"], hp["s_pnums"], hp["t_pnums"])

    graphed = getattr(model, "_graphed", None) is not None
    _g = getattr(model, "_graphed", None)
    model._graphed_kind = ("dp" if type(_g).__name__ == "GraphedStepDP" else
                           "dp-whole" if getattr(_g, "dp", False) else "single")
    from pygda_amd import ops as _ops
    _ops.aggregated_edges = 0
    _ops.kstep_paths = {}
    sync()
    if not graphed:
        profiler.start()
    t0 = time.perf_counter()
    model._train_epochs(*state, epochs=range(args.warmup, total_epochs))
    sync()
    dt = time.perf_counter() - t0
    profiler.stop()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    sustained = None
    if graphed and world == 1 and not side and not args.profile_run and not args.no_sustained and hasattr(_g, "launch"):
        # the 20-step timed region above is 9 ms; the same replays over >= 400 launches, with the device time of every
        # replay (HIP events between consecutive replay ends): what a long fit() sustains, and how it spreads
        from pygda_amd import hipgraph as _hg
        rec, per = [], max(1, getattr(_g, "unroll", 1) if getattr(_g, "graph_multi", None) is not None else 1)
        orig_multi, orig_single = _hg.GraphedStep.launch_multi, _hg.GraphedStep.launch

        def _traced(fn):
            def run(self):
                h0 = time.perf_counter()
                t = fn(self)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                rec.append((h0, time.perf_counter(), ev))
                return t
            return run

        _hg.GraphedStep.launch_multi, _hg.GraphedStep.launch = _traced(orig_multi), _traced(orig_single)
        try:
            n_sus = 400 * per
            sync()
            s0 = time.perf_counter()
            model._train_epochs(*state, epochs=range(10 ** 6, 10 ** 6 + n_sus))
            sync()
            sdt = time.perf_counter() - s0
        finally:
            _hg.GraphedStep.launch_multi, _hg.GraphedStep.launch = orig_multi, orig_single
        dev_ms = sorted(rec[i - 1][2].elapsed_time(rec[i][2]) / per for i in range(1, len(rec)))
        host_us = sorted(1e6 * (b - a) for a, b, _ in rec)
        pick = lambda v, q: v[min(len(v) - 1, int(q * len(v)))]
        sustained = {"replays": len(rec), "steps": n_sus, "steps_per_replay": per, "ms_per_step": 1e3 * sdt / n_sus,
                     "device_ms_per_step_p50": pick(dev_ms, 0.5), "device_ms_per_step_p99": pick(dev_ms, 0.99),
                     "device_ms_per_step_max": dev_ms[-1], "host_launch_call_us_p50": pick(host_us, 0.5),
                     "host_launch_call_us_p99": pick(host_us, 0.99),
                     "replays_slower_than_1p5x_median": sum(1 for v in dev_ms if v > 1.5 * pick(dev_ms, 0.5)),
                     "note": "same captured step, continued after the timed region; per-replay device time from HIP events "
                             "between consecutive replay ends, divided by the steps per replay"}
    execution = (({"dp": "four hipGraph segments with eager RCCL collectives between them",
                   "dp-whole": "hipGraph replay of the whole data-parallel step, RCCL collectives captured "
                               "(library-owned communicator)"}
                  .get(getattr(model, "_graphed_kind", "single"), "hipGraph replay of the captured step"))
                 if graphed else "eager launches")
    def executed_edges():
        """Aggregations actually launched per step x nnz.  The bookkeeping (ops.aggregated_edges, ops.kstep_paths)
        runs with the profiler: eager steps inside the timed region, or the eager pass after a replayed one."""
        if _ops.aggregated_edges == 0:          # captured steps, no eager pass yet: one bookkeeping step
            model.use_hip_graph, model._graphed = False, None
            profiler.start()
            model._train_epochs(*state, epochs=range(total_epochs, total_epochs + 1))
            sync()
            profiler.stop()
            return _ops.aggregated_edges
        return _ops.aggregated_edges // args.steps

    if side:
        executed = executed_edges()
        if rank != 0:
            return None
        return {"what": f"cfg-A as {world} full-batch REPLICAS (one per GPU: cfg-A has one batch per epoch, SURVEY "
                        "8e 'replicas only'), independent dropout draws, global-batch MMD over all-gathered sample "
                        "rows, one flat RCCL gradient all-reduce per step.  The job finishes epochs at "
                        "epochs_per_sec whatever N is: this is the cost of the exchange steps, not a speed-up",
                "ms_per_step": 1e3 * dt / args.steps, "epochs_per_sec": args.steps / dt,
                "edges_aggregated_per_sec_per_replica": executed * args.steps / dt, "replicas": world,
                "execution": execution, "graph": args.graph}
    if args.profile_run:
        if rank != 0:
            return None
        return {"metric": "edges_aggregated_per_sec", "value": edges * args.steps / dt, "unit": "edges/s", "n_gpus": world,
                "value_counts": "reference-equivalent edges (the executed count needs the eager bookkeeping pass this run skips)",
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "profile_run": True,
                "execution": execution, "graph": args.graph}
    host_launch = None
    if graphed and hasattr(_g, "_refill") and hasattr(_g, "_replay"):
        # what the host pays per step: the refill (sample draws + H2D enqueue) and the hipGraphLaunch call itself,
        # timed with an idle device after the timed region (replaying a forked graph enqueues its nodes one by one)
        refill, replay = [], []
        for _ in range(10):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            _g._refill()
            h1 = time.perf_counter()
            _g._replay()
            h2 = time.perf_counter()
            refill.append(h1 - h0)
            replay.append(h2 - h1)
        torch.cuda.synchronize()
        host_launch = {"refill_us": 1e6 * sorted(refill)[5], "graph_launch_call_us": 1e6 * sorted(replay)[5]}
    if graphed:
        # Per-kernel HIP-event timing cannot bracket launches inside a replayed graph, so the
        # roofline inputs are measured on the SAME kernels in an eager pass of the same K steps
        # right after the timed region (same stream, same data, same launch configuration).
        model.use_hip_graph, model._graphed = False, None
        # one UNTIMED eager step first: an eager step meets one-off work a replayed step never does (hub-row layouts
        # and workspaces built at first use, the allocator's first touch of eager-only temporaries); bracketed, that
        # read as a 0.93 ms "kernel" inside a 0.52 ms step in the round-3 driver line
        model._train_epochs(*state, epochs=range(total_epochs, total_epochs + 1))
        sync()
        profile_attempts = 0
        for attempt in range(2):
            profile_attempts += 1
            profiler.start()
            model._train_epochs(*state, epochs=range(total_epochs + 1 + attempt * args.steps,
                                                     total_epochs + 1 + (attempt + 1) * args.steps))
            sync()
            profiler.stop()
            prof_try = profiler.summary()
            # no kernel family can take longer per step than the step (eager steps are slower than replayed ones, so
            # the bound is the eager step's own duration; families on parallel streams each obey it separately)
            if all(v["ms"] / args.steps <= 1e3 * dt / args.steps for v in prof_try.values()):
                break

    prof = profiler.summary()
    executed = executed_edges()
    kstep_paths = {k: v // args.steps for k, v in (getattr(_ops, "kstep_paths", {}) or {}).items()}
    if rank != 0:
        return None
    ms = 1e3 * dt / args.steps

    def kstep_alone():
        """The one-launch K-step kernel of the target graph, 30 launches back to back between two HIP events on the
        launch stream: its duration without the event pair and the neighbours of the per-launch bracket above
        (rocprofv3's in-graph average of the same kernel is the figure to compare with)."""
        try:
            g = as_graph(tgt_d.edge_index, tgt_d.num_nodes)
            g.static = True                      # the full-batch graph of this run (the loader tags its own copy)
            hit = _ops.lds_kstep_plan(g, hp["t_pnums"], False)
            if hit is None:
                return None
            plan, slots = hit
            n, d = g.num_nodes, hp["hid"]
            n_pad = (n + 3) // 4 * 4
            xT = torch.randn(d, n_pad, device=dev)
            yT = torch.empty_like(xT)
            from pygda_amd import _lib as _L
            out = []
            for K_ in (hp["t_pnums"], 0):      # K = 0: plan load + column load / store only (the launch's fixed cost)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for rep in range(33):
                    if rep == 3:
                        e0.record()
                    _L.check(_L.lib().gda_kstep_lds_colmajor_f32(_L.ptr(plan), slots, n, d, K_, _L.ptr(xT), n_pad, _L.ptr(yT),
                                                                 n_pad, None, None, _L.stream()), "gda_kstep_lds_colmajor_f32")
                e1.record()
                torch.cuda.synchronize()
                out.append(1e3 * e0.elapsed_time(e1) / 30)
            return tuple(out)
        except Exception:                     # noqa: BLE001 -- a side figure must not cost the line
            return None, None

    alone_us, fixed_us = kstep_alone()

    # committed PMC passes of this same command (profiles/, made by tools/profile_r3.sh)
    prof_pattern = "r[0-9]*_cfgA" + ("_powerlaw" if args.graph == "powerlaw" else "") + "_rocprof_summary.json"

    def roof(name):
        r = prof[name]
        secs = r["ms"] * 1e-3
        if name.startswith("kstep_lds") and r.get("lds_bytes"):
            # The one-launch K-step kernel keeps its operands in LDS for all K steps: its bound is the LDS gather
            # rate of the CUs it occupies (one workgroup = one CU per feature column), NOT HBM (counter traffic
            # 8 MB per launch = 0.04 of the roof).  `achieved` = the 4-byte words its step loops gather out of LDS
            # (slot-program entries, padding included, x columns x K) over the launch's duration; `peak` = the
            # ds_read_b32 rate of the occupied CUs (128 B/clk/CU x 2.4 GHz, MI355X_MICROARCH.md LDS table).  The
            # duration is the committed rocprofv3 in-graph average of this same command when there is one
            # (`avg_launch_us_rocprof`: the kernel as the replayed step runs it, beside the other branches'
            # kernels); the live figures stand beside it: the eager pass's HIP-event bracket per launch
            # (`avg_launch_us`, event pair included) and 30 launches back to back between two events.
            cols = int(name.split("d=")[1].split(",")[0])
            cus = min(cols, 256)
            peak = LDS_READ_B32_PEAK_GBS * cus / 256.0
            per_launch = r["lds_bytes"] / r["launches"]
            rp_us, rp_calls, rp_file = rocprof_kernel("k_kstep_lds<", prof_pattern)
            # THIS run's durations price the line: the HIP-event bracket of every launch of the eager
            # pass (event pair included: the conservative figure) is `frac`; 30 launches back to back between two events
            # stand beside it.  The committed rocprofv3 in-graph average is a labelled side figure, flagged stale when
            # the summary was made with another build of the library.
            dur_us = r["avg_us"]
            ach = per_launch / (dur_us * 1e-6) / 1e9
            rp_lib = summary_library(rp_file) if rp_file else None
            out = {"kernel": name, "bound": "lds", "achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak,
                   "frac_is": "LDS gather rate over the ds_read_b32 rate of the CUs the launch occupies "
                              f"({cus} workgroups = {cus} of 256 CUs)",
                   "duration_used_us": dur_us,
                   "duration_source": "HIP events of THIS run: per-launch bracket on the launch stream in the eager pass "
                                      "after the timed region (same kernels, same data, same launch configuration)",
                   "avg_launch_us": r["avg_us"], "back_to_back_launch_us": alone_us,
                   "frac_back_to_back": (per_launch / (alone_us * 1e-6) / 1e9 / peak) if alone_us else None,
                   "rocprof_committed": None if not rp_us else {
                       "avg_launch_us": rp_us, "launches": rp_calls, "file": rp_file, "what": "rocprofv3 in-graph average",
                       "frac": per_launch / (rp_us * 1e-6) / 1e9 / peak, "library_sha16": rp_lib,
                       "same_library_as_this_run": rp_lib is not None and rp_lib == library_sha16()},
                   "lds_bytes_gathered_per_launch": per_launch, "launches": r["launches"],
                   "cus_occupied": cus, "frac_of_whole_chip": ach / LDS_READ_B32_PEAK_GBS}
            # ... of which useful: one 4-byte word per stored entry of the graph, column and step (the plan pads its slots and
            # splits hub rows: `frac` counts those reads too) -- VERDICT round 4, item 2
            K_name = int(name.split("K=")[1].rstrip("]"))
            useful = 4.0 * nnz_t * cols * K_name
            out["useful_gather_bytes_per_launch"] = useful
            out["useful_gather_frac"] = useful / (dur_us * 1e-6) / 1e9 / peak
            out["gathered_words_that_are_padding"] = 1.0 - useful / per_launch if per_launch else None
            if alone_us and fixed_us and alone_us > fixed_us:
                # where the launch's time goes: K = 0 launches cost `fixed_us` (the slot program, 262 KB per workgroup,
                # and the column in / out); the rest is the K step loops, whose LDS gather rate is the kernel's bound
                out["back_to_back_launch_us_K0"] = fixed_us
                out["frac_in_step_loops_back_to_back"] = per_launch / ((alone_us - fixed_us) * 1e-6) / 1e9 / peak
            out["traffic"], out["traffic_source"] = pmc_traffic("k_kstep_lds", prof_pattern)
            out["traffic_provenance"] = traffic_provenance(out["traffic_source"])
            if r.get("hbm_bytes"):
                out["hbm_bytes_per_launch"] = r["hbm_bytes"] / r["launches"]
                out["hbm_frac_real"] = r["hbm_bytes"] / r["launches"] / (dur_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            # SURVEY 8(d)'s accounting for comparison only: the K aggregations one launch stands for, as if each had
            # moved its algorithmic bytes over HBM -- an equivalence, not a bandwidth
            alg = r["bytes"] / r["launches"]
            out["frac_survey_8d"] = alg / (dur_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            out["frac_survey_8d_is"] = ("SURVEY 8(d): the algorithmic bytes of the K aggregations one launch stands for "
                                        "(K x (nnz*8 + (N+1)*4 + 2*N*d*4)) over this launch's duration over 8 TB/s -- an "
                                        "equivalence, the operands never leave LDS")
            if alone_us:
                out["frac_survey_8d_back_to_back"] = alg / (alone_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            out["algorithmic_equivalent"] = {"bytes_per_launch": alg, "K_aggregations_per_launch": int(name.split("K=")[1].rstrip("]")),
                                             "GBs": alg / (dur_us * 1e-6) / 1e9,
                                             "algorithmic_equivalent_frac": alg / (dur_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
            return out
        if name.startswith("spmm") or name.startswith("kstep_lds"):
            ach = r["bytes"] / secs / 1e9
            out = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": ach / HBM_PEAK_GBS, "launches": r["launches"],
                   "avg_launch_us": r["avg_us"], "algorithmic_bytes_per_launch": r["bytes"] / r["launches"]}
            out["traffic"], out["traffic_source"] = pmc_traffic("k_spmm<32, 4", prof_pattern) if "d=128" in name else (None, None)
            if "d=128" in name:
                rp_us, rp_calls, rp_file = rocprof_kernel("k_spmm<32, 4, false, false, false>", prof_pattern)
                if rp_us:
                    out["avg_launch_us_rocprof"], out["rocprof_source"] = rp_us, rp_file
                    out["frac_rocprof"] = r["bytes"] / r["launches"] / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            out["note"] = ("cache-resident at this size (the whole operand set fits the Infinity Cache): the HBM "
                           "fraction is SURVEY 8(d)'s accounting, see roofline_hbm_regime for the HBM-bound size")
            return out
        ach = r["flops"] / secs / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": ach / FP32_MFMA_PEAK_TF, "traffic": None, "launches": r["launches"],
                "avg_launch_us": r["avg_us"]}

    cands = [k for k in prof if k.startswith("spmm") or k.startswith("kstep_lds") or k.startswith("dense")]
    dominant = max(cands, key=lambda k: prof[k]["ms"])
    agg = max((k for k in prof if k.startswith("spmm") or k.startswith("kstep_lds")), key=lambda k: prof[k]["ms"])
    graph_note = ("shape-identical stand-in graphs" if args.graph == "uniform" else
                  "stand-in graphs with the same N / E and Zipf (power-law) degrees")
    out = {
        "metric": "edges_aggregated_per_sec", "value": executed * args.steps / dt, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "dtype_note": ("fp32 values everywhere; the MMD's two pair products run as three fp16 MFMAs on split operands (hi + lo, 22 "
                       "significant bits, fp32 accumulation) when the one-pass kernel covers the shape -- error against float64 "
                       "equal to the fp32-MFMA kernels' (profiles/r4_mmd_one_pass_experiments.txt); PYGDA_AMD_MMD_ONE_PASS=0 "
                       "runs those instead"),
        "data": "synthetic",
        "config": {"workload": f"cfg-A: A2GNN ACMv9->DBLPv7 ({graph_note}, "
                               "Ns=9360/Es=15556, Nt=5484/Et=8117, F=6775), nhid=128, L=2, s_pnums=0, "
                               "t_pnums=10, weight=10, dropout=0.5, full batch, "
                               + ("adversarial" if args.adv else "MMD") + " domain loss",
                   "edges_aggregated_per_step": executed,
                   "edges_aggregated_per_step_reference_equivalent": edges,
                   "note": "value counts the aggregations EXECUTED; layer 0 is evaluated once per domain and "
                           "shared by the two passes the reference runs separately (identical values), so a "
                           "step executes fewer aggregations than the reference's step",
                   "nnz_source": nnz_s, "nnz_target": nnz_t, "graph": args.graph,
                   "max_row_source": max_row(as_graph(src_d.edge_index, src_d.num_nodes)),
                   "max_row_target": max_row(as_graph(tgt_d.edge_index, tgt_d.num_nodes)),
                   "execution": execution, "kstep_aggregation_path": kstep_paths,
                   "parallelism": "single GPU" if world == 1 else f"{world} replicas"},
        "epochs_per_sec": args.steps / dt,
        "reference_equivalent_edges_per_sec": edges * args.steps / dt,
        "roofline": dict(roof(dominant), timing="HIP events on the launch stream, " + (
            "eager pass of the same K steps after the timed hipGraph region" if graphed else "timed region")),
        "roofline_aggregation": roof(agg),
        # BASELINE.json also asks for the MFMA utilisation of the dense projection: the hidden-layer
        # products on the hand-written matrix-core kernels (layer 0 is a sparse projection here)
        "roofline_dense_projection": {k: roof(k) for k in sorted(prof) if k.startswith("dense_projection")},
        "kernel_time_ms_per_step": {k: v["ms"] / args.steps for k, v in sorted(prof.items())},
    }
    # no kernel family can take longer per step than the step: an entry that still does after the warm eager step and one
    # retry is an event-bracket artefact and is reported as such, not as a kernel time
    bad = {k: v for k, v in out["kernel_time_ms_per_step"].items() if v > ms}
    if bad:
        out["kernel_time_anomalies"] = bad
        for k in bad:
            del out["kernel_time_ms_per_step"][k]
    assert all(v <= ms for v in out["kernel_time_ms_per_step"].values())
    if not args.adv and "mmd_fwd" in prof:
        # the MMD pair (profiles/HISTORY.md 4.3): the two matrix-core kernels against the products they evaluate.  times x [m x m] pair
        # tiles over d features: the distance product (only the upper triangle of the symmetric matrix is computed) and
        # the backward's G x T product (full).  Durations: committed rocprofv3 averages per kernel when present; the live
        # brackets cover the whole C call (forward = 4 kernels, backward = 2)
        times, m_rows, dd = 5, 2000, hp["hid"]
        full = 2.0 * m_rows * m_rows * dd * times
        from pygda_amd import ops as _ops
        one_pass = _ops.mmd_one_pass_segments(times, m_rows // 2, dd) > 0
        per_call = lambda k: prof.get(k, {}).get("ms", 0.0) * 1e3 / max(prof.get(k, {}).get("calls", 1), 1)
        if one_pass:
            # ONE pass over the pairs (csrc/gda_mmd_fused.inc): distance product + gradient product, the full matrix each
            # (2 x `full`), on the 16-bit matrix cores with split operands -- three fp16 MFMAs per fp32-equivalent product.
            # `frac` prices the MFMAs it executes against the fp16 roof; `fp32_equivalent` prices the product it
            # stands for against the fp32-MFMA roof the two-pass kernels run under (it may exceed 1: that is the point).
            mm = {"shape": {"times": times, "m": m_rows, "d": dd}, "path": "one pass, split-fp16 MFMA (3 products per pair)",
                  "bound": "mfma", "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s", "flops_full_product": full,
                  "live_call_us": {"mmd_fwd[tile_split+fused+finalize]": per_call("mmd_fwd"),
                                   "mmd_bwd[scatter]": per_call("mmd_bwd")}}
            us, calls, fname = rocprof_kernel("k_mmd_fused<", prof_pattern)
            if us:
                mm["rocprof_same_library_as_this_run"] = summary_library(fname) is not None and summary_library(fname) == library_sha16()
                mm["k_mmd_fused"] = {"avg_launch_us_rocprof": us, "source": fname, "flops_executed": 3 * 2 * full,
                                     "achieved": 3 * 2 * full / (us * 1e-6) / 1e12,
                                     "frac": 3 * 2 * full / (us * 1e-6) / 1e12 / F16_MFMA_PEAK_TF,
                                     "fp32_equivalent": {"flops": 2 * full, "achieved": 2 * full / (us * 1e-6) / 1e12,
                                                         "frac_of_fp32_mfma_peak": 2 * full / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF}}
            for key, prefix in (("k_tile_split", "k_tile_split<"), ("k_finalize", "k_finalize"), ("k_bwd_scatter", "k_bwd_scatter<")):
                us, calls, fname = rocprof_kernel(prefix, prof_pattern)
                if us:
                    mm[key] = {"avg_launch_us_rocprof": us, "source": fname}
        else:
            mm = {"shape": {"times": times, "m": m_rows, "d": dd}, "path": "two passes, fp32 MFMA", "bound": "mfma",
                  "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "flops_full_product": full,
                  "live_call_us": {"mmd_fwd[rowstats+bandwidth+pairdist+finalize]": per_call("mmd_fwd"),
                                   "mmd_bwd[k_bwd+scatter]": per_call("mmd_bwd")}}
            for key, prefix, flops in (("k_pairdist", "k_pairdist<", full * (m_rows / 64 + 1) / (2 * m_rows / 64)),
                                       ("k_bwd", "k_bwd<", full)):
                us, calls, fname = rocprof_kernel(prefix, prof_pattern)
                if us:
                    mm[key] = {"avg_launch_us_rocprof": us, "source": fname, "flops_executed": flops,
                               "achieved": flops / (us * 1e-6) / 1e12, "frac": flops / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF}
        out["roofline_mmd"] = mm
    if host_launch is not None:
        out["host_per_step"] = host_launch
    if sustained is not None:
        out["sustained"] = sustained
    out["library_sha16"] = library_sha16()
    if graphed:
        out["profile_pass_attempts"] = profile_attempts       # 2: the first eager pass met an event-bracket artefact and was repeated
    # the trainer and its captured graphs form a reference cycle: collect it HERE, device idle -- left to the cyclic
    # collector, the hipGraphs (and their memory pool) were destroyed whenever it next ran, e.g. inside the timed
    # region of the cfg-S side line that follows (a 100 ms stall in a 33 ms region)
    del model, state
    import gc
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    if world == 1 and not args.no_hbm_probe:
        # cfg-A's graphs are cache resident, so the HBM fraction above says little about the kernel:
        # the same aggregation kernel timed at configs[4]'s per-domain size, where HBM is the bound
        out["roofline_hbm_regime"] = hbm_regime_probe(dev)
    if world == 1 and not args.no_cpu_baseline:
        # LAST thing of the process (main() calls it after the side line): ~27 s of all host cores, whose after-effects
        # -- spinning worker threads, page-cache and allocator churn -- the host-bound cfg-S side line should not meet
        out["_cpu_baseline_thunk"] = lambda: cpu_baseline(src, tgt, hp, edges)
    return out


def other_configs(dev, epochs=30, which=("grade_mmd", "grade_js", "udagcn", "adagcn")):
    """BASELINE.json configs[2] / configs[3] in the measured set: GRADE Citationv1 -> DBLPv7 with
    the MMD and the (script default) JS discriminator, UDAGCN and AdaGCN ACMv9 -> Citationv1, full batch, at the
    hyper-parameters of benchmark/node/run_citation.sh:45,109-110 on shape-identical stand-ins (Citationv1: 8,935 nodes /
    15,098 edges).  Per trainer: steady-state ms/epoch of ``fit()`` as it runs by default (the captured step where the
    trainer has one), then a short eager pass under the HIP-event profiler for the dominant kernel family and its roofline."""
    import numpy as np
    import pygda_amd
    from pygda_amd import profiler
    M = pygda_amd.models
    pairs = {"C->D": make_cfg_a(seed=204, ns=8935, es=15098, nt=5484, et=8117),
             "A->C": make_cfg_a(seed=203, ns=9360, es=15556, nt=8935, et=15098)}
    F = 6775

    def build(name, graphed, ep):
        kw = dict(device=dev, epoch=ep, verbose=0, use_hip_graph=graphed)
        if name == "grade_mmd":
            return "C->D", M.GRADE(F, 128, 5, num_layers=5, dropout=0.5, disc="MMD", weight=0.01, lr=0.001, weight_decay=0.001, **kw)
        if name == "grade_js":
            return "C->D", M.GRADE(F, 128, 5, num_layers=5, dropout=0.5, disc="JS", weight=0.01, lr=0.001, weight_decay=0.001, **kw)
        if name == "udagcn":
            return "A->C", M.UDAGCN(F, 128, 5, num_layers=2, ppmi=True, adv_dim=40, lr=0.0001, weight_decay=0.001, **kw)
        return "A->C", M.AdaGCN(F, 128, 5, num_layers=2, dropout=0.4, adv_dim=40, lr=0.01, weight_decay=0.01, **kw)

    import gc
    out = {}
    for name in which:
        try:
            torch.manual_seed(0)
            np.random.seed(0)
            task, m = build(name, None, epochs)
            src, tgt = pairs[task]
            stamps = []
            m.epoch_hook = lambda e, loss, acc, secs: stamps.append((time.perf_counter(), loss))
            m.fit(src, tgt)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            half = len(stamps) // 2           # steady state: the first epochs hold ingestion, PPMI builds, the capture
            ms = (t1 - stamps[half - 1][0]) / (len(stamps) - half) * 1e3
            res = {"task": task, "ms_per_epoch": ms, "epochs": epochs,
                   "execution": "hipGraph replay" if getattr(m, "_graphed", None) is not None else "eager launches",
                   "finite": bool(np.isfinite(stamps[-1][1]))}
            del m
            torch.cuda.synchronize()
            gc.collect()
            # the kernel families of one eager epoch
            task, m = build(name, False, 4)
            seen = []
            m.epoch_hook = lambda e, loss, acc, secs: (seen.append(e), profiler.start() if e == 0 else None)
            m.fit(src, tgt)
            torch.cuda.synchronize()
            profiler.stop()
            prof = profiler.summary()
            n_ep = max(len(seen) - 1, 1)
            if prof:
                dom = max(prof, key=lambda k: prof[k]["ms"])
                r = prof[dom]
                secs = r["ms"] * 1e-3
                hbm = dom.startswith("spmm") or dom.startswith("kstep") or dom.startswith("sparse")
                from pygda_amd import ops as _ops
                # the one-pass MMD kernels execute split-fp16 MFMAs (three products per pair: the flops the region counts)
                f16 = dom.startswith("mmd") and _ops.MMD_ONE_PASS
                res["dominant_kernel"] = {
                    "kernel": dom + (" [tile split + one-pass pair kernel + finalize, split-fp16 MFMAs]" if f16 else ""),
                    "ms_per_epoch": r["ms"] / n_ep, "launches_per_epoch": r["launches"] / n_ep,
                    "bound": "hbm" if hbm else "mfma",
                    "frac": (r["bytes"] / secs / 1e9 / HBM_PEAK_GBS) if hbm
                    else (r["flops"] / secs / 1e12 / (F16_MFMA_PEAK_TF if f16 else FP32_MFMA_PEAK_TF)),
                    "frac_is": ("SURVEY 8(d) algorithmic bytes over the HIP-event duration over 8 TB/s (cache-resident at this "
                                "size: an accounting figure)" if hbm else
                                "executed fp16 MFMA flops over the HIP-event duration of the whole call over the fp16 MFMA peak" if f16
                                else "flops over the HIP-event duration over the fp32 MFMA peak"),
                    "timing": "HIP events around the family's launches, eager epochs"}
                res["kernel_time_ms_per_epoch"] = {k: v["ms"] / n_ep for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:6]}
            out[name] = res
            del m
            torch.cuda.synchronize()
            gc.collect()
        except Exception as exc:                  # noqa: BLE001 -- a side object must not cost the line
            out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            profiler.stop()
    torch.cuda.empty_cache()
    return out


def side_line_child(side_args):
    """``python bench.py --workload cfgS`` (same step / warm-up counts, same graph options) in a child process; its full
    result object, or None."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "cfgS", "--steps", str(side_args.steps), "--warmup",
           str(side_args.warmup), "--nodes", str(side_args.nodes), "--avg-degree", str(side_args.avg_degree), "--feat",
           str(side_args.feat), "--batch", str(side_args.batch), "--fanout", side_args.fanout, "--no-cpu-baseline",
           "--no-strict-fp32", "--full-line"] + (["--eager"] if side_args.eager else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return d if "ms_per_step" in d and "config" in d else None
    except Exception:                 # noqa: BLE001
        return None


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, budget=1990):
    """The contract line, under 2 kB: the driver's keys, `roofline` (with SURVEY 8(d)'s fraction inside it), `cpu_baseline`,
    `kernel_time_ms_per_step` and one short object per side measurement.  Everything else -- the long explanatory strings,
    the per-kernel rooflines, the sustained / HBM-regime / MMD objects -- is in the details object (stderr line and
    bench_details.json), which this line names."""
    c = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "epochs_per_sec", "steps_per_sec", "reference_equivalent_edges_per_sec",
              "library_sha16", "profile_run", "functional_check", "rccl_ranks_seen")
    c["vs_baseline"] = out.get("vs_baseline")                  # null: BASELINE.md holds no published number for this metric
    cfg = out.get("config", {})
    c["config"] = _pick(cfg, "edges_aggregated_per_step", "edges_aggregated_per_step_reference_equivalent", "graph",
                        "host_ms_per_step_max_median", "host_work_ms_per_step", "hipMalloc_calls_in_timed_region",
                        "aggregation_launches_per_step")
    c["config"]["workload"] = str(cfg.get("workload", ""))[:150]
    for k in ("execution", "parallelism"):
        if cfg.get(k):
            c["config"][k] = str(cfg[k])[:90]
    r = out.get("roofline") or {}
    c["roofline"] = _pick(r, "kernel", "bound", "achieved", "peak", "unit", "frac", "frac_survey_8d", "traffic",
                          "avg_launch_us", "back_to_back_launch_us", "back_to_back_launch_us_K0",
                          "gathered_words_that_are_padding", "launches", "cus_occupied")
    if "frac_is" in r:
        c["roofline"]["frac_is"] = str(r["frac_is"])[:110]
    if "cpu_baseline" in out:
        c["cpu_baseline"] = _pick(out["cpu_baseline"], "value", "unit", "cores", "kind", "cpu")
        smp = str(out["cpu_baseline"].get("sample", ""))          # the line keeps a short form, never shed (details: in full)
        c["cpu_baseline"]["sample"] = (smp.split(" (")[0] + " after 1 warm-up step:" + smp.rsplit(":", 1)[-1])[:80] if ":" in smp else smp[:80]
    if "kernel_time_ms_per_step" in out:
        c["kernel_time_ms_per_step"] = {k.replace("dense_projection", "gemm").replace("sparse_projection", "spgemm"): round(v, 4)
                                        for k, v in out["kernel_time_ms_per_step"].items()}
    sr = out.get("scaling_reference")
    if sr:
        c["scaling_reference"] = _pick(sr, "value", "unit", "ms_per_step", "steps", "host_work_ms_per_step",
                                       "host_ms_per_step_max_median", "reference_equivalent_edges_per_sec")
    if out.get("sustained"):
        c["sustained"] = _pick(out["sustained"], "ms_per_step", "device_ms_per_step_p50", "device_ms_per_step_p99", "steps")
    hb = out.get("roofline_hbm_regime")
    if hb:
        c["hbm_regime"] = _pick(hb, "frac", "traffic_over_algorithmic", "avg_launch_us")
        rm = (hb.get("rmat_2^22") or {}).get("trainer_default")
        if rm:
            c["hbm_regime"]["rmat"] = _pick(rm, "frac", "traffic_over_algorithmic")
    mm = (out.get("roofline_mmd") or {}).get("k_mmd_fused")
    if mm:
        c["mmd_fused"] = _pick(mm, "avg_launch_us_rocprof", "frac")
    if out.get("strict_fp32"):
        c["strict_fp32_ms_per_step"] = {k: (v.get("ms_per_step") if isinstance(v, dict) else None)
                                        for k, v in out["strict_fp32"].items()}
    if out.get("other_configs"):
        c["other_configs_ms_per_epoch"] = {k: (round(v["ms_per_epoch"], 4) if isinstance(v, dict) and "ms_per_epoch" in v else None)
                                           for k, v in out["other_configs"].items()}
    if out.get("cfgA_replicas"):
        c["cfgA_replicas"] = _pick(out["cfgA_replicas"], "ms_per_step", "epochs_per_sec", "replicas", "error")
    c["details"] = "bench_details.json + the stderr line: all objects in full"

    def rnd(v):                                 # six significant digits are what a reader compares; 17 are what json prints
        if isinstance(v, float):
            return float(f"{v:.6g}")
        if isinstance(v, dict):
            return {k: rnd(x) for k, x in v.items()}
        if isinstance(v, list):
            return [rnd(x) for x in v]
        return v
    c = rnd(c)
    # the driver's tail is 2,000 characters: shed the least-read parts first, never the contract keys / roofline / cpu_baseline
    size = lambda: len(json.dumps(c))
    if size() > budget and "kernel_time_ms_per_step" in c:
        c["kernel_time_ms_per_step"] = dict(sorted(c["kernel_time_ms_per_step"].items(), key=lambda kv: -kv[1])[:8])
    for k in ("roofline.frac_is", "config.execution", "config.workload", "mmd_fused", "sustained",
              "hbm_regime", "other_configs_ms_per_epoch", "kernel_time_ms_per_step"):
        if size() <= budget:
            break
        a, _, b = k.partition(".")
        if b:
            if k == "config.workload":
                c["config"]["workload"] = c["config"]["workload"][:60]
            else:
                c.get(a, {}).pop(b, None)
        else:
            c.pop(a, None)
    return c


def emit(out, args):
    """stdout: ONE line -- the compact contract line (or, with --full-line, everything).  The whole object also goes to
    stderr (one line, ahead of the contract line) and to bench_details.json beside this script (gpurun_out/ when it exists)."""
    full = json.dumps(out)
    if args.full_line:
        print(full)
        return
    try:
        d = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(d if os.path.isdir(d) else ROOT, "bench_details.json")
        with open(path, "w") as fh:
            fh.write(full + "\n")
    except OSError:
        pass
    sys.stderr.write(full + "\n")
    sys.stderr.flush()
    print(json.dumps(compact_line(out)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 timed steps by default (76 ms of cfg-A): a 20-step window is five replays, and what surrounds them -- the first
    # refill and launch before the device has anything to do, the last read-back -- cost it 0.45 ms, 6 % of the window
    # (20 steps 0.401 ms/step, the same replays sustained over 1600 steps 0.379: profiles/r6_bench_details.json)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true",
                    help="skip the 5 M-node / 100 M-edge aggregation timed after the cfg-A region (roofline_hbm_regime)")
    ap.add_argument("--adv", action="store_true", help="adversarial branch instead of MMD")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a hipGraph")
    ap.add_argument("--workload", default=None, choices=["cfgA", "cfgS"],
                    help="cfgA: BASELINE.json configs[1] (default at --gpus 1); cfgS: sampled mini-batches on "
                         "configs[4]-style graphs, seed shards per rank (default at --gpus N > 1, where cfg-A -- one "
                         "batch per epoch -- can only run as replicas: SURVEY 8e)")
    ap.add_argument("--graph", default="uniform", choices=["uniform", "powerlaw"],
                    help="cfgA stand-in graphs: uniform random pairs, or the same N / E with Zipf degrees (hubs of "
                         "several hundred neighbours, as real citation graphs have)")
    ap.add_argument("--nodes", type=int, default=5_000_000, help="cfgS: nodes per domain")
    ap.add_argument("--avg-degree", type=int, default=20)
    ap.add_argument("--feat", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1024, help="cfgS: seeds per GPU per step")
    ap.add_argument("--fanout", default="15,10")
    ap.add_argument("--no-side-lines", action="store_true",
                    help="skip the second workload's labelled side object (N = 1: cfg-S on one GPU, the base point "
                         "of the scaling curve; N > 1: cfg-A replicas)")
    ap.add_argument("--side-steps", type=int, default=60,
                    help="timed steps of the labelled side line (60: one rare 30 - 50 ms host stall, profiles/HISTORY.md 5, moves a 30-step\n"
                         "figure by 1.1 - 1.6 ms/step)")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel code path (RCCL exchange steps) on a 1-rank group")
    ap.add_argument("--profile-run", action="store_true",
                    help="for runs under rocprofv3: stop after the timed region (no eager HIP-event pass, no back-to-back "
                         "kernel probes), so that the profiler's per-kernel averages are those of the replayed steps")
    ap.add_argument("--no-sustained", action="store_true", help="skip the 400-replay sustained measurement of cfg-A")
    ap.add_argument("--no-strict-fp32", action="store_true",
                    help="skip the strict-fp32 companion figures (child processes with PYGDA_AMD_MMD_ONE_PASS=0 and "
                         "PYGDA_AMD_GEMM_SPLIT_F16=0)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the configs[2] / configs[3] side object (GRADE, UDAGCN, AdaGCN epoch times)")
    ap.add_argument("--side-in-process", action="store_true",
                    help="measure the N = 1 cfg-S scaling reference inside this process instead of a child process")
    ap.add_argument("--full-line", action="store_true",
                    help="print the whole result object as the stdout line (default: the compact contract line on stdout, "
                         "the whole object on stderr and in bench_details.json)")
    ap.add_argument("--no-rccl-direct", action="store_true",
                    help="collectives through torch.distributed's ProcessGroup instead of the library-owned RCCL communicator")
    ap.add_argument("--rccl-direct", action="store_true",
                    help="collectives through the C ABI's own RCCL communicator (gda_allreduce_f32 / "
                         "gda_allgather_f32): the whole data-parallel step is then ONE hipGraph")
    ap.add_argument("--share-gpus", action="store_true",
                    help="FUNCTIONAL check of the N-rank path on a node with fewer than N GPUs: rank r takes GPU "
                         "r mod (GPUs visible), the group is gloo (RCCL refuses two ranks on one device).  The line "
                         "says so (`functional_check`) and is not a measurement")
    ap.add_argument("--launch-check", action="store_true",
                    help="bring up the --gpus ranks, check the group's size with one all-reduce, print it and exit "
                         "(gloo when the box has no GPU: the CPU test of the launcher)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    import pygda_amd                      # first of all: caps the host thread pool at the container's CPU quota (pygda_amd/_cpu.py)
    from pygda_amd import _cpu
    throttle0 = _cpu.throttle_counters()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(relaunch(args))          # start the ranks ourselves; each re-enters main() with the env set
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}")
    gpu = torch.cuda.is_available()
    if not gpu and not args.launch_check:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if gpu:
        if args.share_gpus:
            local %= torch.cuda.device_count()
        if torch.cuda.device_count() <= local:
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local}, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    import torch.distributed as dist
    if world > 1 or args.force_dp:
        init_group(args, world, rank, dev)
    if args.launch_check:
        if rank == 0:
            print(json.dumps({"launch_check": "ok", "n_gpus": dist.get_world_size() if dist.is_initialized() else 1,
                              "backend": dist.get_backend() if dist.is_initialized() else None}))
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    workload = args.workload or ("cfgA" if world == 1 else "cfgS")
    side_args = argparse.Namespace(**vars(args))
    side_args.steps = args.side_steps if args.steps >= 20 else min(args.steps, args.side_steps)
    side_args.warmup = max(args.warmup, 10) if args.steps >= 20 else min(args.warmup, 5)     # a fresh allocator pool: see profiles/HISTORY.md 5
    if workload == "cfgS":
        out = run_cfg_s(args, world, rank, dev)
        if world > 1 and not args.no_side_lines:
            # The side object runs code that has never met two GPUs (segmented / whole-step capture with collectives
            # between the ranks): an exception is caught below, a HANG cannot be -- so a watchdog per rank ends the
            # process cleanly after PYGDA_AMD_BENCH_SIDE_TIMEOUT seconds, rank 0 printing the scaling line it already
            # holds with the side object marked as timed out.
            import threading
            limit = float(os.environ.get("PYGDA_AMD_BENCH_SIDE_TIMEOUT", "240"))

            def give_up():
                if rank == 0:
                    out["cfgA_replicas"] = {"error": f"watchdog: the cfg-A replicas side measurement did not finish within "
                                                     f"{limit:.0f} s and was abandoned; the scaling line above it is complete"}
                    if args.share_gpus:
                        out["functional_check"] = f"{world} ranks over gloo (--share-gpus): NOT a measurement"
                    emit(out, args)
                    sys.stdout.flush()
                sys.stderr.write(f"bench.py rank {rank}: side-object watchdog fired after {limit:.0f} s\n")
                sys.stderr.flush()
                os._exit(0)

            guard = threading.Timer(limit, give_up)
            guard.daemon = True
            guard.start()
            try:                          # a labelled side object must not cost the scaling line
                side = run_cfg_a(side_args, world, rank, dev, side=True)
            except Exception as exc:      # noqa: BLE001 -- reported in the line, rank-local
                side = {"error": f"{type(exc).__name__}: {exc}"[:400]}
            finally:
                guard.cancel()
            if rank == 0:
                out["cfgA_replicas"] = side
    else:
        out = run_cfg_a(args, world, rank, dev)
        if world == 1 and not args.no_side_lines and not args.force_dp:
            # the base point of the 1/2/4/8 scaling curve: the sampled workload the N > 1 lines report, on one GPU
            # measured in a CHILD process, as the N > 1 lines it is the reference of are (fresh process, nothing of the
            # cfg-A phase -- its captured graphs' pools, streams, helper threads -- beside it: in-process the same steps
            # ran 2.24 ms against 1.93 standalone in round 6); in-process only if the child fails
            side = None if args.side_in_process else side_line_child(side_args)
            if side is None:
                side = run_cfg_s(side_args, world, rank, dev, cpu_base=False)
                side["measured"] = "in this process, after the cfg-A phase"
            out["scaling_reference"] = {
                "measured": side.get("measured", "child process (python bench.py --workload cfgS ...), like the N > 1 lines"),
                "what": "cfg-S on this one GPU: `bench.py --gpus N` (N > 1) reports cfg-S (seed shards per rank, "
                        "weak scaling); divide its value by N x this value for the scaling efficiency",
                "value": side["value"], "unit": side["unit"], "ms_per_step": side["ms_per_step"],
                "steps": side["steps"], "workload": side["config"]["workload"],
                "sampler": side["config"].get("sampler"),
                # the host side of those steps: [slowest, median] step, and where the slowest one spent its time -- a
                # single slow step is 1/30 of this figure
                "host_ms_per_step_max_median": side["config"].get("host_ms_per_step_max_median"),
                "host_work_ms_per_step": side["config"].get("host_work_ms_per_step"),
                "host_wait_ms_per_step": side["config"].get("host_wait_ms_per_step"),
                "hipMalloc_calls_in_timed_region": side["config"].get("hipMalloc_calls_in_timed_region"),
                "execution": side["config"].get("execution"),
                "edges_aggregated_per_step": side["config"].get("edges_aggregated_per_step"),
                "edges_aggregated_per_step_reference_equivalent": side["config"].get("edges_aggregated_per_step_reference_equivalent"),
                "reference_equivalent_edges_per_sec": side.get("reference_equivalent_edges_per_sec"),
                "host_slowest_step": (side["config"].get("host_phases") or {}).get("slowest_step"),
                "roofline": side["roofline"], "roofline_dense_projection": side["roofline_dense_projection"]}
        elif world > 1 and rank == 0:
            out["config"]["parallelism"] = (f"{world} full-batch replicas (explicit --workload cfgA; value is ONE "
                                            "replica's rate, the job's epoch rate is epochs_per_sec)")
        thunk = out.pop("_cpu_baseline_thunk", None) if isinstance(out, dict) else None
        if (rank == 0 and world == 1 and not args.no_other_configs and not args.profile_run and not args.force_dp
                and not args.adv and not args.eager and not args.no_side_lines):
            out["other_configs"] = other_configs(dev)
        strict = (rank == 0 and world == 1 and not args.no_strict_fp32 and not args.profile_run and not args.force_dp
                  and not args.adv and not args.eager)
        if strict:
            # BEFORE the CPU baseline (30 s of all host cores): the children want a quiet host
            out["strict_fp32"] = {"cfgA": strict_fp32_companion("cfgA", args.steps, args.warmup, ["--graph", args.graph])}
            if "scaling_reference" in out:
                out["strict_fp32"]["cfgS"] = strict_fp32_companion("cfgS", min(args.steps, 30), min(max(args.warmup, 5), 10))
        if thunk is not None:
            out["cpu_baseline"] = thunk()
    if rank == 0:
        throttle1 = _cpu.throttle_counters()
        out["host_cpu"] = {"intra_op_threads": torch.get_num_threads(), "cgroup_quota_cores": _cpu.cpu_quota(),
                           "visible_cpus": len(os.sched_getaffinity(0)),
                           "cgroup_throttle_events_during_this_process": None if throttle0 is None or throttle1 is None
                           else throttle1[0] - throttle0[0],
                           "note": "a throttle event freezes every thread of the container for the rest of a 100 ms "
                                   "period: pygda_amd/_cpu.py keeps the intra-op pool inside the quota"}
        if getattr(args, "_ranks_seen", None) is not None:
            out["rccl_ranks_seen"] = args._ranks_seen            # all-gather of the ranks over the group (init_group)
            out["collectives"] = args._collectives
        if args.share_gpus:
            out["functional_check"] = (f"{world} ranks on {torch.cuda.device_count()} GPU(s) over gloo (--share-gpus): "
                                       "the N-rank code path end to end, NOT a measurement")
        emit(out, args)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
