"""CPU: the C-ABI library loads and exports every symbol include/gda_hip.h declares; the
host-side mirror of the reference interface (ctor validation, Data, loaders, metrics,
logger) behaves like the reference.  No kernels are launched here."""
import ctypes
import io
import os
import re
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import pygda_amd
from pygda_amd import _lib
from pygda_amd.data import Data, NeighborLoader, to_undirected
from pygda_amd.metrics import eval_macro_f1, eval_micro_f1
from pygda_amd.models import A2GNN, GRADE, BaseGDA
from pygda_amd.utils import logger

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gda_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gda_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in gda_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert L.gda_abi_version() == 1
    assert L.gda_status_string(0) == b"ok"
    assert b"workspace" in L.gda_status_string(-3)
    assert L.gda_mmd_workspace_bytes(5, 1000, 128) > 0
    assert L.gda_graph_workspace_bytes(1000, 100) > 0


def test_argument_validation_without_gpu():
    """Status codes for bad arguments come back before anything touches a device."""
    L = _lib.lib()
    assert L.gda_spmm_csr_f32(None, None, None, 10, 4, None, 4, None, 4, None, None) == -1     # NULL
    assert L.gda_spmm_csr_f32(None, None, None, -1, 4, None, 4, None, 4, None, None) == -2     # size
    assert L.gda_spmm_csr_kstep_f32(None, None, None, 10, 4, 0, None, 4, None, 4, None, None, None) == -2
    assert L.gda_build_csr_norm(None, None, None, 5, 3, 1.0, 1, 1, 0, None, None, None, None, None, None,
                                None, 0, None) == -1
    assert L.gda_mmd_fwd_f32(None, 4, None, 4, 4, None, None, 1, 8, 2.0, 5, 0.0, None, None, None, None, 0,
                             None) == -1
    assert L.gda_gather_rows_f32(None, 4, 8, None, 3, None, 8, None) == -2                    # ldx < d


def test_round3_entry_points_validate_before_touching_a_device():
    """The entry points added in round 3: capacity maths of the device sampler (host-only), envelopes of the tall /
    skinny GEMMs and of the interior-rows K-step, argument errors of the graph-mode readout."""
    import ctypes
    import numpy as np
    L = _lib.lib()
    one = ctypes.c_void_p(1)                                   # non-NULL dummies: validation comes first
    fan = np.array([15, 10], dtype=np.int32)
    nc, ec = ctypes.c_int64(), ctypes.c_int64()
    assert L.gda_dsampler_caps(1024, fan.ctypes.data, 2, 400, 100_000_000, 5_000_000, ctypes.byref(nc), ctypes.byref(ec)) == 0
    assert (nc.value, ec.value) == (1024 * (1 + 15 + 150), 1024 * (15 + 150))      # seeds x (1 + k1 + k1 k2), picks
    assert L.gda_dsampler_workspace_bytes(1024, fan.ctypes.data, 2, 400, 100_000_000, 5_000_000) > 0
    small = np.array([15, 10], dtype=np.int32)
    assert L.gda_dsampler_caps(64, small.ctypes.data, 2, 3, 500, 200, ctypes.byref(nc), ctypes.byref(ec)) == 0
    assert nc.value == 200 and ec.value == 64 * 3 + 500       # bounded by the graph: in-degree <= 3, E = 500 edges (in-lists are disjoint), N = 200 nodes
    for bad in ([0, 5], [65], [4, 0]):                         # fan-out 0 or above 64: the host sampler's business
        b = np.array(bad, dtype=np.int32)
        assert L.gda_dsampler_caps(8, b.ctypes.data, len(bad), 10, 100, 50, ctypes.byref(nc), ctypes.byref(ec)) == -4
        assert L.gda_dsampler_workspace_bytes(8, b.ctypes.data, len(bad), 10, 100, 50) == 0
    allk = np.array([-1, -1], dtype=np.int32)                  # whole neighbourhoods: bounded by the edge count
    assert L.gda_dsampler_caps(8, allk.ctypes.data, 2, 10, 100, 50, ctypes.byref(nc), ctypes.byref(ec)) == 0
    assert nc.value == 50 and ec.value == 80 + 100
    assert L.gda_dsampler_sample(None, None, 10, 5, 2, None, 1, fan.ctypes.data, 2, 0, None, None, None, None, None, None,
                                 None, None, None, None, None, 0, None) == -1
    assert L.gda_dsampler_build_graph(one, one, 2 ** 31, 10, one, one, one, one, 1 << 40, None) == -2      # int32 edge count
    # the one-call batch of the recycling loader and its events: arguments are checked before anything is enqueued
    args = [None] * 30
    args[2:5], args[6], args[9], args[10], args[23], args[28] = [10, 5, 2], 1, 2, 0, 0, 0
    assert L.gda_dsampler_batch(*args) == -1                                   # no counts / landing pad
    args[20], args[24] = one, one
    assert L.gda_dsampler_batch(*args) == -1                                   # seeds without a device copy
    args[5], args[7], args[6] = one, one, -1
    assert L.gda_dsampler_batch(*args) == -2                                   # negative seed count
    args[6], args[21] = 1, one
    assert L.gda_dsampler_batch(*args) == -1                                   # one plan without the other
    for fn in (L.gda_event_record, L.gda_stream_wait_event):
        assert fn(None, None) == -1
    assert L.gda_event_synchronize(None) == -1 and L.gda_event_create(None) == -1 and L.gda_event_destroy(None) == 0
    # GEMM envelopes: refused shapes are the general kernel's business
    assert L.gda_gemm_tall_f32(0, 100_000, 96, 128, one, 128, one, 128, ctypes.c_void_p(2), 96, None, None, None, 0, None) == -4
    assert L.gda_gemm_tall_f32(0, 100_000, 128, 100, one, 100, one, 100, ctypes.c_void_p(2), 128, None, None, None, 0, None) == -4
    assert L.gda_gemm_tall_f32(2, 64, 128, 100_000, one, 64, one, 128, ctypes.c_void_p(2), 128, None, None, None, 0, None) == -4
    assert L.gda_gemm_tall_workspace_bytes(2, 128, 256, 150_000) == 256 * 128 * 257 * 4
    assert L.gda_gemm_tall_workspace_bytes(0, 150_000, 128, 256) == 0
    assert L.gda_gemm_skinny_f32(0, 100_000, 9, 128, one, 128, one, 128, ctypes.c_void_p(2), 9, None, None, None, 0, None) == -4
    assert L.gda_gemm_skinny_f32(0, 100_000, 5, 100, one, 100, one, 100, ctypes.c_void_p(2), 5, None, None, None, 0, None) == -4
    assert L.gda_gemm_skinny_f32(2, 5, 128, 100_000, one, 5, ctypes.c_void_p(16), 128, ctypes.c_void_p(2), 128, None, None,
                                 None, 0, None) == -3          # the slab partials need a workspace
    assert L.gda_gemm_skinny_workspace_bytes(2, 5, 128, 100_000) == min(512, 100_000 // 64) * 5 * 129 * 4   # <= 512 row slabs
    # interior-rows K-step / graph-mode readout
    assert L.gda_spmm_csr_interior_kstep_f32(one, one, one, 10, 11, 4, 2, 0, one, ctypes.c_void_p(2), one, None, None, None) == -2
    assert L.gda_spmm_csr_interior_kstep_f32(one, one, one, 10, 4, 4, 2, 0, one, ctypes.c_void_p(2), None, None, None, None) == -1
    assert L.gda_spmm_csr_interior_kstep_f32(one, one, one, 10, 4, 4, 0, 0, one, ctypes.c_void_p(2), one, None, None, None) == -2
    # transposed-output step: the leading dimension of yT counts ROWS
    assert L.gda_spmm_csr_tout_f32(one, one, one, 10, 4, one, 4, ctypes.c_void_p(2), 9, None, None) == -2
    assert L.gda_spmm_csr_tout_f32(one, one, one, 10, 4, one, 3, ctypes.c_void_p(2), 12, None, None) == -2
    assert L.gda_spmm_csr_tout_f32(None, one, one, 10, 4, one, 4, ctypes.c_void_p(2), 12, None, None) == -1
    assert L.gda_spmm_csr_tout_f32(one, one, one, 0, 4, one, 4, ctypes.c_void_p(2), 12, None, None) == 0      # nothing to do
    assert L.gda_segment_mean_fwd_f32(None, 4, None, 3, 4, None, 4, None) == -1
    assert L.gda_segment_mean_fwd_f32(one, 2, one, 3, 4, one, 4, None) == -2                   # ldx < d
    assert L.gda_segment_mean_bwd_f32(one, 4, one, None, 5, 4, one, 4, None) == -1


def test_graph_mode_collation_and_loader_on_the_host():
    """PyG's Batch.from_data_list for what graph mode reads, and torch's own DataLoader underneath: the shuffles are
    the installed torch's draws from the default generator (a2gnn.py:278-286 relies on exactly that)."""
    import torch.utils.data as tud
    from pygda_amd.data import DataLoader, collate_graphs
    from oracle import pygda_cpu as O
    g = torch.Generator().manual_seed(4)
    ds = [Data(x=torch.randn(3 + i, 4, generator=g), edge_index=torch.randint(0, 3 + i, (2, 5 + i), generator=g),
               y=torch.tensor([i % 3])) for i in range(7)]
    b = collate_graphs(ds)
    ob = O.collate_graphs([O.Graph(d.x, d.edge_index, d.y) for d in ds])
    for k in ("x", "edge_index", "y", "batch"):
        assert torch.equal(getattr(b, k), getattr(ob, k))
    assert b.num_graphs == 7 and b.batch.tolist() == sum(([i] * (3 + i) for i in range(7)), [])
    assert int(b.edge_index[:, 5:11].min()) >= 3            # the second graph's edges are shifted by the first's nodes
    torch.manual_seed(3)
    mine = [bb.y.tolist() for bb in DataLoader(ds, batch_size=3, shuffle=True)]
    torch.manual_seed(3)
    ref = [bb.y.tolist() for bb in tud.DataLoader(ds, batch_size=3, shuffle=True, collate_fn=collate_graphs)]
    assert mine == ref and [len(m) for m in mine] == [3, 3, 1]
    one = next(iter(DataLoader(ds, batch_size=7)))
    assert one.batch._gda_sorted and one.batch._gda_num_graphs == 7 and one.y.tolist() == [0, 1, 2, 0, 1, 2, 0]


def test_bench_reads_the_pmc_summary_of_its_round(tmp_path, monkeypatch):
    """bench.py's `traffic` comes from the committed PMC passes: the newest ROUND's summary of the same command."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    for rnd, val in (("r2", 1.0), ("r10", 3.0), ("r3", 2.0)):
        (prof / f"{rnd}_cfgA_rocprof_summary.json").write_text(json.dumps({"pmc": {"k_kstep_lds<8>(x)": {"traffic_bytes": val}}}))
    (prof / "r11_cfgA_powerlaw_rocprof_summary.json").write_text(json.dumps({"pmc": {"k_kstep_lds<10>(x)": {"traffic_bytes": 9.0}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("k_kstep_lds", "r[0-9]*_cfgA_rocprof_summary.json") == (3.0, "r10_cfgA_rocprof_summary.json")
    assert bench.pmc_traffic("k_kstep_lds", "r[0-9]*_cfgA_powerlaw_rocprof_summary.json")[0] == 9.0
    assert bench.pmc_traffic("k_nothing", "r[0-9]*_cfgA_rocprof_summary.json") == (None, None)


def test_fit_freezes_the_collector_once(monkeypatch):
    """models/base.py::_freeze_gc: one collect + freeze per process (a generation-2 pass over the interpreter and torch
    is tens of milliseconds = many sampled steps), left alone when PYGDA_AMD_GC_FREEZE=0."""
    import gc
    from pygda_amd.models import base as B
    monkeypatch.setattr(B, "_gc_frozen", False)
    monkeypatch.setenv("PYGDA_AMD_GC_FREEZE", "0")
    before = gc.get_freeze_count()
    B._freeze_gc()
    assert gc.get_freeze_count() == before and B._gc_frozen is False
    monkeypatch.setenv("PYGDA_AMD_GC_FREEZE", "1")
    try:
        B._freeze_gc()
        frozen = gc.get_freeze_count()
        assert frozen > before and B._gc_frozen is True
        junk = [[] for _ in range(1000)]
        B._freeze_gc()                                     # second call: nothing more is frozen
        assert gc.get_freeze_count() == frozen and len(junk) == 1000
    finally:
        gc.unfreeze()                                      # the test process goes on collecting as usual


def test_loss_terms_is_a_plain_tuple_of_its_terms():
    from pygda_amd import hipgraph
    a, b = torch.tensor(1.5, requires_grad=True), torch.tensor(2.0, requires_grad=True)
    terms = hipgraph.LossTerms((a * 2, b * 3))
    assert isinstance(terms, tuple) and len(terms) == 2 and hipgraph.defer_total is False
    torch.autograd.backward(list(terms), [torch.ones(()) for _ in terms])      # what GraphedStep does with them
    assert float(a.grad) == 2.0 and float(b.grad) == 3.0


def test_product_path_has_no_cpu_fallback():
    x = torch.randn(4, 8)
    ei = torch.tensor([[0, 1], [1, 0]])
    conv = pygda_amd.nn.PropGCNConv(8, 4)
    with pytest.raises(_lib.GdaError):
        conv(x, ei, 2)
    with pytest.raises(_lib.GdaError):
        pygda_amd.utils.get_MMD(torch.randn(4, 3), torch.randn(4, 3))


def test_round4_entry_points_validate_before_touching_a_device():
    """gda_step_bump / gda_adam_multi_ex_f32 (round 4: the step-counter bump at the start of a step)."""
    import ctypes
    L = _lib.lib()
    one, two = ctypes.c_void_p(8), ctypes.c_void_p(16)
    assert L.gda_step_bump(None, None, 0, None) == 0                               # nothing to do
    assert L.gda_step_bump(None, None, 3, None) == -1                              # NULL table
    assert L.gda_step_bump(None, None, -1, None) == -2 and L.gda_step_bump(None, None, 49, None) == -2
    arr = (ctypes.c_void_p * 2)(one, one)
    assert L.gda_step_bump(None, arr, 2, None) == -4          # the same counter twice
    arr = (ctypes.c_void_p * 2)(one, None)
    assert L.gda_step_bump(None, arr, 2, None) == -1
    # fused view attention (csrc/gda_attention.hip)
    two = (ctypes.c_void_p * 2)(ctypes.c_void_p(16), ctypes.c_void_p(32))
    ld = (ctypes.c_int64 * 2)(128, 128)
    assert L.gda_attention_workspace_bytes(9360, 128) == ((9360 + 7) // 8) * 129 * 4
    assert L.gda_attention_fuse_fwd_f32(1, two, ld, 10, 128, one, one, one, 128, one, None) == -4          # one view
    assert L.gda_attention_fuse_fwd_f32(2, two, ld, 10, 130, one, one, one, 130, one, None) == -4          # h % 4
    assert L.gda_attention_fuse_fwd_f32(2, two, ld, 10, 516, one, one, one, 516, one, None) == -4          # h > 512
    assert L.gda_attention_fuse_fwd_f32(2, None, ld, 10, 128, one, one, one, 128, one, None) == -1
    assert L.gda_attention_fuse_fwd_f32(2, two, ld, 10, 128, None, one, one, 128, one, None) == -1
    assert L.gda_attention_fuse_fwd_f32(2, two, ld, 0, 128, None, None, None, 128, None, None) == 0        # no rows
    assert L.gda_attention_fuse_bwd_f32(2, two, ld, 10, 128, two, two, two, 128, two, two, two, None, 0, None) == -3   # workspace
    # one-pass MMD (csrc/gda_mmd_fused.inc): what it covers, and its argument checks
    assert L.gda_mmd_fused_nseg(5, 1000, 128, 2.0, 5) == 6          # 16 column blocks x 5 resamples x 6 = 480 workgroups
    assert L.gda_mmd_fused_nseg(1, 96, 128, 2.0, 5) == 6            # never more segments than 32-row tiles
    assert L.gda_mmd_fused_nseg(3, 200, 64, 2.0, 5) == 8
    for bad in ((5, 1000, 645, 2.0, 5), (5, 1000, 160, 2.0, 5), (5, 1000, 16, 2.0, 5), (5, 1000, 128, 3.0, 5),
                (5, 1000, 128, 2.0, 4), (0, 1000, 128, 2.0, 5), (5, 0, 128, 2.0, 5)):
        assert L.gda_mmd_fused_nseg(*bad) == 0
    one, two = ctypes.c_void_p(64), ctypes.c_void_p(128)            # "pointers" on 16-byte boundaries, never dereferenced
    fwd = lambda **kw: L.gda_mmd_fused_fwd_f32(*[kw.get(k, v) for k, v in (
        ("src", one), ("lds", 128), ("tgt", two), ("ldt", 128), ("d", 128), ("si", None), ("ti", None), ("times", 5),
        ("n", 1000), ("mul", 2.0), ("num", 5), ("sigma", 0.0), ("scale", 1.0), ("add", None), ("rs", None), ("rt", None),
        ("loss", one), ("bw", one), ("part", two), ("nseg", 6), ("ws", one), ("wsb", 0), ("stream", None))])
    assert fwd() == -3                                               # workspace too small: the last check before a launch
    assert fwd(src=None) == -1 and fwd(loss=None) == -1 and fwd(part=None) == -1 and fwd(ws=None) == -1
    assert fwd(nseg=4) == -2                                         # not the segment count of these shapes
    assert fwd(d=645, lds=645, ldt=645) == -4 and fwd(num=4) == -4 and fwd(mul=3.0) == -4
    assert fwd(si=one, ti=two) == -4                                 # indexed rows need the gathered copy
    assert fwd(si=one) == -1                                         # one index without the other
    assert fwd(src=ctypes.c_void_p(4)) == -4 and fwd(part=ctypes.c_void_p(8)) == -4      # 16-byte alignment
    assert fwd(lds=130) == -4                                        # row stride off the 16-byte grid
    bwd = L.gda_mmd_fused_bwd_f32
    assert bwd(None, 6, 5, 1000, 128, one, 1.0, one, None, None, 0, None, None, None, 0, None, None) == -1
    assert bwd(one, 6, 5, 1000, 128, None, 1.0, one, None, None, 0, None, None, None, 0, None, None) == -1
    assert bwd(one, 9, 5, 1000, 128, one, 1.0, one, None, None, 0, None, None, None, 0, None, None) == -2
    assert bwd(one, 6, 5, 1000, 128, one, 1.0, None, None, None, 0, None, None, None, 0, None, None) == -1   # nowhere to write
    assert bwd(one, 6, 5, 1000, 128, one, 1.0, None, one, None, 0, one, one, one, 0, one, None) == -1        # half a selection CSR
    assert bwd(one, 6, 5, 1000, 128, one, 1.0, None, one, one, 0, one, one, one, 0, one, None) == 0          # no feature rows
    assert L.gda_mmd_workspace_bytes(5, 1000, 128) > L.gda_mmd_workspace_bytes(5, 1000, 132) - 5 * 63 * 38912   # the images
    table = (_lib.AdamTensorStruct * 1)()
    # a second gradient contribution summed inside the update (gda_adam_multi_sum_f32)
    full = (_lib.AdamTensorStruct * 1)(_lib.AdamTensorStruct(64, 128, 192, 256, 320, 10))
    assert L.gda_adam_multi_sum_f32(full, (ctypes.c_void_p * 1)(128), 1, 0.1, 0.9, 0.999, 1e-8, 0.0, 0, None) == -5   # the gradient twice
    assert L.gda_adam_multi_sum_f32(full, (ctypes.c_void_p * 1)(64), 1, 0.1, 0.9, 0.999, 1e-8, 0.0, 0, None) == -5    # the parameter itself
    assert L.gda_adam_multi_sum_f32(None, None, 1, 0.1, 0.9, 0.999, 1e-8, 0.0, 0, None) == -1
    assert L.gda_adam_multi_sum_f32(full, None, 1, 0.1, 0.9, 0.999, 1e-8, 0.0, 4, None) == -4
    assert L.gda_adam_multi_ex_f32(table, 1, 0.1, 0.9, 0.999, 1e-8, 0.0, 2, None) == -4   # flag
    assert L.gda_adam_multi_ex_f32(table, 0, 0.1, 0.9, 0.999, 1e-8, 0.0, 1, None) == 0
    assert L.gda_adam_multi_ex_f32(None, 2, 0.1, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1


def test_basegda_num_neigh_validation():
    m = A2GNN(8, 4, 3, num_layers=2, num_neigh=[15, 10], device="cpu")
    assert m.num_neigh == [15, 10]
    assert A2GNN(8, 4, 3, num_layers=3, device="cpu").num_neigh == [-1, -1, -1]
    with pytest.raises(ValueError):
        A2GNN(8, 4, 3, num_layers=2, num_neigh=[15], device="cpu")
    with pytest.raises(ValueError):
        GRADE(8, 4, 3, num_layers=2, num_neigh="all", device="cpu")
    with pytest.raises(TypeError):
        BaseGDA(8, 4, 3)          # abstract


def test_reference_signatures_and_defaults():
    import inspect
    sig = inspect.signature(A2GNN.__init__).parameters
    want = dict(mode='node', num_layers=3, dropout=0., s_pnums=0, t_pnums=30, adv=False, weight=5,
                weight_decay=0., lr=4e-3, epoch=200, device='cuda:0', batch_size=0, num_neigh=-1, verbose=2)
    for k, v in want.items():
        assert sig[k].default == v, k
    assert sig['act'].default is F.relu
    g = inspect.signature(GRADE.__init__).parameters
    assert (g['disc'].default, g['weight'].default, g['weight_decay'].default, g['lr'].default) == ('JS', 0.01, 0.01, 0.001)
    p = inspect.signature(pygda_amd.nn.PropGCNConv.forward).parameters
    assert list(p)[1:] == ['x', 'edge_index', 'prop_nums', 'edge_weight'] and p['prop_nums'].default == 1
    c = inspect.signature(pygda_amd.nn.CachedGCNConv.forward).parameters
    assert c['cache_name'].default == "default_cache"
    mm = inspect.signature(pygda_amd.utils.MMD).parameters
    assert (mm['sampling_num'].default, mm['times'].default) == (1000, 5)


def test_init_rng_stream_matches_reference_golden():
    """Same seed -> same initial weights as the reference's A2GNNBase (double glorot draw
    per conv, torch default init for the discriminator)."""
    from tests.conftest import load_golden, sub
    for adv in (False, True):
        g = load_golden("a2gnn_forward_adv" if adv else "a2gnn_forward_mmd")
        torch.manual_seed(int(g["init_seed"]))
        net = pygda_amd.nn.A2GNNBase(24, 16, 5, num_layers=2, adv=adv, dropout=0.0)
        sd = net.state_dict()
        assert set(sd) == set(sub(g, "param/"))
        for k, v in sub(g, "param/").items():
            np.testing.assert_array_equal(sd[k].numpy(), v)
    g = load_golden("grade_forward_js")
    torch.manual_seed(int(g["init_seed"]))
    net = pygda_amd.nn.GRADEBase(24, 8, 5, num_layers=3, dropout=0.0, disc="JS")
    for k, v in sub(g, "param/").items():
        np.testing.assert_array_equal(net.state_dict()[k].numpy(), v)


def test_data_and_full_batch_loader():
    x = torch.randn(6, 3)
    ei = torch.tensor([[0, 1, 2, 2], [1, 0, 3, 3]])
    d = Data(x=x, edge_index=ei, y=torch.arange(6))
    assert d.num_nodes == 6 and d.num_edges == 4 and d.num_node_features == 3
    assert d.to("cpu") is d
    loader = NeighborLoader(d, [-1, -1], batch_size=6)
    assert len(loader) == 1 and next(iter(loader)) is d
    und = to_undirected(ei, 6)
    assert und.size(1) == 4      # (0,1),(1,0),(2,3),(3,2); the duplicate is merged
    assert Data(x=x, edge_index=und).is_undirected() and not d.is_undirected()


def test_metrics_match_sklearn():
    from sklearn.metrics import f1_score
    g = torch.Generator().manual_seed(0)
    for c in (2, 5, 9):
        y = torch.randint(0, c, (500,), generator=g)
        p = torch.randint(0, c, (500,), generator=g)
        assert abs(eval_micro_f1(y, p) - f1_score(y.numpy(), p.numpy(), average="micro")) < 1e-12
        assert abs(eval_macro_f1(y, p) - f1_score(y.numpy(), p.numpy(), average="macro")) < 1e-12
    # a class that is predicted but absent from the labels still counts (sklearn semantics)
    y, p = torch.tensor([0, 0, 1, 1]), torch.tensor([0, 2, 1, 1])
    assert abs(eval_macro_f1(y, p) - f1_score(y.numpy(), p.numpy(), average="macro")) < 1e-12


def test_logger_format():
    buf = io.StringIO()
    with redirect_stdout(buf):
        logger(epoch=3, loss=1.23456, source_train_acc=0.5, time=2.0, verbose=2, train=True)
        logger(epoch=3, loss=1.0, verbose=0)
        logger(epoch=4, loss=(1.0, 2.0), verbose=1)
    lines = buf.getvalue().splitlines()
    assert lines[0] == "Epoch 0003: loss 1.2346, source acc 0.5000, time 2.00"
    assert lines[1] == "Epoch 0004: Loss I 1.0000 | Loss O 2.0000 | "


def test_citation_dataset_reader(tmp_path):
    """The three-text-file format of pygda/datasets/citation.py:153-173, CRLF labels included."""
    from pygda_amd.datasets import CitationDataset
    raw = tmp_path / "raw"
    raw.mkdir()
    (raw / "toy_edgelist.txt").write_text("0,1\n1,2\n2,0\n3,3\n")
    (raw / "toy_docs.txt").write_text("0,1,0.5\n1,0,0\n0,0,1\n1,1,1\n")
    (raw / "toy_labels.txt").write_bytes(b"0\r\n2\r\n1\r\n2\r\n")
    ds = CitationDataset(str(tmp_path), "toy")
    d = ds[0]
    assert d.edge_index.dtype == torch.int64 and d.edge_index.tolist() == [[0, 1, 2, 3], [1, 2, 0, 3]]
    assert d.x.dtype == torch.float32 and d.x.tolist() == [[0, 1, 0.5], [1, 0, 0], [0, 0, 1], [1, 1, 1]]
    assert d.y.dtype == torch.int64 and d.y.tolist() == [0, 2, 1, 2]
    assert ds.num_classes == 3 and ds.num_node_features == 3 and len(ds) == 1
    m = d.train_mask.int() + d.val_mask.int() + d.test_mask.int()
    assert m.tolist() == [1, 1, 1, 1] and int(d.train_mask.sum()) == 3
    again = CitationDataset(str(tmp_path), "toy")[0]                 # served from the binary cache
    assert torch.equal(again.train_mask, d.train_mask) and torch.equal(again.x, d.x)
    with pytest.raises(FileNotFoundError):
        CitationDataset(str(tmp_path / "nope"), "toy")


def test_pygda_alias_package():
    """Scripts written against the reference import ``pygda``; the alias resolves to this build."""
    import pygda
    from pygda.models import A2GNN as A, GRADE, UDAGCN, AdaGCN, DANE, GNN
    from pygda.nn import PropGCNConv, CachedGCNConv, GradReverse, A2GNNBase
    from pygda.utils import MMD, logger as lg
    from pygda.metrics import eval_micro_f1 as f1
    from pygda.datasets import CitationDataset
    from pygda.nn.prop_gcn_conv import gcn_norm
    assert A is A2GNN and PropGCNConv is pygda_amd.nn.PropGCNConv and MMD is pygda_amd.utils.MMD
    assert f1 is eval_micro_f1 and lg is logger and gcn_norm is pygda_amd.nn.gcn_norm


# ------------------------------------------------------------------------- TDSS host side --
def test_tdss_khop_builder_matches_reference_golden():
    """gda_two_hop_host + self-loop pass == the reference's spspmm/coalesce smoothing graphs."""
    from pygda_amd.models import TDSS
    from tests.conftest import load_golden, T
    g = load_golden("tdss")
    ei, nt = T(g["tgt_ei"]), g["tgt_x"].shape[0]
    for k in (1, 2, 3):
        m = TDSS(24, 16, 5, smooth_mode='K-hop', k=k, device="cpu")
        got, attr = m.smoothness(ei, None, nt)
        assert attr is None
        assert np.array_equal(got.numpy(), g[f"khop{k}_ei"])


def test_tdss_walk_builder_properties():
    """RW smoothing graph: own generator, so structural checks -- sorted unique (visited, start)
    pairs, every start paired with itself, every pair reachable within rw_len steps, and the
    per-start visit counts distributed like the stub walker's."""
    from pygda_amd.models.tdss import walk_smooth_edges
    from oracle import pygda_cpu as O
    from tests.conftest import load_golden, T
    g = load_golden("tdss")
    ei, nt = T(g["tgt_ei"]), g["tgt_x"].shape[0]
    got = walk_smooth_edges(ei, nt, 4, seed=5).numpy()
    key = got[0] * nt + got[1]
    assert np.all(np.diff(key) > 0)                                   # sorted by (row, col), unique
    pairs = set(zip(got[0].tolist(), got[1].tolist()))
    assert all((i, i) in pairs for i in range(nt))
    reach = O.tdss_smoothness_khop(ei, nt, 3)                         # <= 4 hops (two squarings) + loops
    allowed = set(zip(reach[1].tolist(), reach[0].tolist()))          # (visited, start): start -> visited
    assert pairs <= allowed
    again = walk_smooth_edges(ei, nt, 4, seed=5).numpy()
    assert np.array_equal(got, again)                                 # reproducible per seed
    other = walk_smooth_edges(ei, nt, 4, seed=6).numpy()
    assert other.shape != got.shape or not np.array_equal(other, got)
    ref_n = g["rw_ei"].shape[1]                                       # one draw of the reference walker
    assert abs(got.shape[1] - ref_n) < 0.15 * ref_n
    # a node without out-edges stays put: only its own loop
    lonely = walk_smooth_edges(torch.tensor([[0], [1]]), 3, 4, seed=1).numpy()
    assert set(zip(lonely[0].tolist(), lonely[1].tolist())) == {(0, 0), (1, 0), (1, 1), (2, 2)}


def test_tdss_signature_and_asserts():
    import inspect
    from pygda_amd.models import TDSS
    sig = inspect.signature(TDSS.__init__)
    want = dict(mode='node', smooth_mode='RW', num_layers=2, dropout=0., s_pnums=0, t_pnums=30, k=2, rw_len=4,
                alpha=0.001, beta=1e-4, weight_decay=0.005, adv=False, lr=0.01, epoch=200, device='cuda:0',
                batch_size=0, num_neigh=-1, verbose=2)
    for name, default in want.items():
        assert sig.parameters[name].default == default, name
    with pytest.raises(AssertionError):
        TDSS(4, 4, 2, mode='graph')
    with pytest.raises(AssertionError):
        TDSS(4, 4, 2, adv=True)


def test_svd_transform_laplacian_and_shapes(tmp_path):
    from pygda_amd.utils.svd_transform import laplacian_dense, svd_transform
    ei = torch.tensor([[0, 1, 1, 2, 2, 2, 0], [1, 0, 2, 1, 2, 0, 1]])            # a loop and a duplicate
    L = laplacian_dense(ei, 4)
    want = np.array([[2, -1, 0, 0], [-1, 2, -1, 0], [-1, -1, 2, 0], [0, 0, 0, 0]], dtype=np.float32)
    assert np.array_equal(L, want)                                               # D - A, loops dropped
    g = torch.Generator().manual_seed(0)
    n = 130
    ei = torch.randint(0, n, (2, 600), generator=g)
    d = Data(x=torch.zeros(n, 3), edge_index=ei, y=torch.zeros(n, dtype=torch.long))
    svd_transform(d, str(tmp_path) + "/")
    assert d.eivec.shape == (100, n) and d.eival.shape == (100,)
    assert torch.equal(torch.load(str(tmp_path) + "/eivec.pt"), d.eivec)
    # principal directions of a symmetric matrix: orthonormal rows
    gram = d.eivec @ d.eivec.t()
    assert torch.allclose(gram, torch.eye(100), atol=1e-3)


def test_strurw_signature_and_mode_guard():
    import inspect
    from pygda_amd.models import StruRW
    sig = inspect.signature(StruRW.__init__).parameters
    want = dict(num_layers=2, cls_dim=128, cls_layers=2, dropout=0., gnn='GS', pooling='mean', reweight=True,
                pseudo=True, ew_start=100, ew_freq=20, lamb=0.8, mode='erm', bn=False, weight_decay=0.0001, lr=0.05,
                epoch=100, device='cuda:0', batch_size=0, num_neigh=-1, verbose=2)
    for k, v in want.items():
        assert sig[k].default == v, k
    with pytest.raises(AssertionError):
        StruRW(4, 4, 2, mode='other')
    assert StruRW(4, 4, 2, mode='mixup', device='cpu').init_model().__class__.__name__ == 'MixupBase'


def _mixup_on_oracle(monkeypatch):
    """The mixup trainer's host logic on CPU tensors: the oracle's scatter-add underneath the conv's graph /
    aggregation calls (the product kernels have no CPU path), everything above them is the product's code."""
    from oracle import pygda_cpu as O
    import pygda_amd.nn.mixup_gcnconv as MC

    def build_csr(edge_index, n, val, add_self_loops=False, normalize=False):
        return edge_index, val

    def propagate(x, graph, K=1, bias=None):
        return O.propagate(graph[0], graph[1], x)

    monkeypatch.setattr(MC, "build_csr", build_csr)
    monkeypatch.setattr(MC, "propagate", propagate)


def test_strurw_mixup_host_logic_against_reference_fit(monkeypatch):
    """fit()/predict() of mode='mixup' against the reference's 3-epoch trajectory: numpy draws in the
    reference's order (lam, then the shuffle), re-weighting on the step that computes it, unit weights put back
    by predict(), ONE aggregation per layer standing in for the reference's three (the shuffled graph is a
    renumbering)."""
    from tests.conftest import load_golden, sub
    from pygda_amd.data import Data
    from pygda_amd.models import StruRW
    _mixup_on_oracle(monkeypatch)
    g = load_golden("strurw_mixup")
    T = torch.from_numpy
    s = Data(x=T(g["src_x"]), edge_index=T(g["src_ei"]), y=T(g["src_y"]))
    t = Data(x=T(g["tgt_x"]), edge_index=T(g["tgt_ei"]), y=T(g["tgt_y"]))
    losses = []
    m = StruRW(12, 8, 3, num_layers=2, dropout=0.0, reweight=True, pseudo=True, ew_start=2, ew_freq=1, lamb=0.8,
               mode="mixup", lr=0.01, weight_decay=0.001, device="cpu", epoch=3, verbose=0)
    m.epoch_hook = lambda e, loss, acc, secs: losses.append((loss, acc))
    torch.manual_seed(int(g["fit_seed"]))
    np.random.seed(int(g["fit_np_seed"]))
    m.fit(s, t)
    np.testing.assert_allclose([l for l, _ in losses], g["fit/losses"], rtol=1e-5)
    np.testing.assert_allclose([a for _, a in losses], g["fit/accs"], rtol=1e-6)
    logits, labels = m.predict(t)
    np.testing.assert_allclose(logits.numpy(), g["fit/tgt_logits"], atol=2e-5)
    assert np.array_equal(labels.numpy(), g["fit/tgt_labels"])
    for k, v in sub(g, "fit/final/").items():
        np.testing.assert_allclose(m.gnn.state_dict()[k].numpy(), v, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("layers", [2, 3])
def test_mixup_base_foreign_edge_index_b(monkeypatch, layers):
    """A materialised ``edge_index_b`` tensor (what a caller of the reference's MixupBase passes) takes the
    second-aggregation route and lands on the reference's numbers, like the ShuffledEdges record does."""
    from tests.conftest import load_golden, sub
    from pygda_amd.nn import MixupBase, ShuffledEdges
    _mixup_on_oracle(monkeypatch)
    g = load_golden("strurw_mixup")
    tag = f"L{layers}"
    T = torch.from_numpy
    x, ei, y = T(g["src_x"]), T(g["src_ei"]), T(g["src_y"])
    w = T(g[f"{tag}/src_edge_weight"])
    perm, lam = g[f"{tag}/perm"], float(g[f"{tag}/lam"])
    torch.manual_seed(int(g["init_seed"]))
    net = MixupBase(12, 8, 3, num_layers=layers, dropout=0.0, rw_lmda=0.8)
    for k, v in sub(g, f"{tag}/param/").items():
        np.testing.assert_array_equal(net.state_dict()[k].numpy(), v, err_msg=k)
    record = ShuffledEdges(ei, perm)
    for edge_index_b in (record, record.tensor()):
        net.zero_grad()
        logits = net(x, ei, edge_index_b, lam, perm, w)
        np.testing.assert_allclose(logits.detach().numpy(), g[f"{tag}/src_logits"], atol=1e-5)
        torch.nn.functional.cross_entropy(logits, y).backward()
        for k, v in sub(g, f"{tag}/grad/").items():
            np.testing.assert_allclose(dict(net.named_parameters())[k].grad.numpy(), v, atol=1e-5, err_msg=k)


def _kstep_plan_views(buf, S, hub=False):
    import numpy as np
    TB, R = 1024, S * 4
    o1 = TB * R * 8
    o2 = o1 + TB * S * 4
    o3 = o2 + TB * 4
    o4 = o3 + (16320 + 2 + 3) // 4 * 4 * 4
    ent = buf[:o1].view(np.int32).reshape(16, R, 64, 2)
    outa = buf[o1:o2].view(np.uint32).reshape(16, S, 64)
    keep = buf[o2:o3].view(np.uint32)
    pos = buf[o3:o4].view(np.uint32)
    if hub:
        return ent, outa, keep, pos, buf[o4:o4 + TB * 8].view(np.uint32).reshape(TB, 2)
    return ent, outa, keep, pos


def _kstep_emulate_step(ent, outa, keep, hubtab, hub_waves, S, cur):
    """One step of the per-lane program on a word array (csrc/gda_kstep.hip: ks_step, then ks_combine on the
    lanes of the first `hub_waves` waves), fp32 operation by operation."""
    import numpy as np
    nxt = np.full(cur.shape, np.nan, dtype=np.float32)
    nxt[0] = 0.0                                                 # each buffer's zero word
    for t in range(1024):
        wv, lane = t >> 6, t & 63
        acc = np.float32(0)
        for s in range(S):
            for l in range(4):
                a, wb = ent[wv, s * 4 + l, lane]
                acc = np.float32(acc + np.float32(np.int32(wb).view(np.float32) * cur[a // 4]))
            nxt[outa[wv, s, lane] // 4] = acc
            if not (keep[t] >> s) & 1:
                acc = np.float32(0)
    for wv in range(hub_waves):
        v = np.array([nxt[hubtab[wv * 64 + l, 0] // 4] for l in range(64)], dtype=np.float32)
        gm = hubtab[wv * 64:(wv + 1) * 64, 1] >> 24
        off = 1
        while off < 64:
            o = np.concatenate([v[off:], v[:off]])               # __shfl_down: lanes past the end keep their own value,
            o[64 - off:] = v[64 - off:]                          # which no group ever uses
            v = np.where((gm & off) != 0, (v + o).astype(np.float32), v)
            off <<= 1
        for l in range(64):
            out = int(hubtab[wv * 64 + l, 1] & 0xffffff)
            if out:
                nxt[out // 4] = v[l]
    return nxt


def _kstep_lds_cycles(ent, outa, pos, n, banks=32):
    """LDS-array cycles of one step + one column load under MI355X_MICROARCH.md's ds_read_b32 / ds_write_b32 model:
    per wave instruction two lane groups of 32, each costing the largest number of DISTINCT addresses on one bank."""
    import numpy as np

    def cost(addr):                                  # [..., 64] byte addresses
        a = addr.reshape(-1, 2, 32) // 4
        tot = 0
        for grp in a.reshape(-1, 32):
            u = np.unique(grp)
            tot += np.bincount(u % banks, minlength=banks).max()
        return int(tot)
    n_pad = (n + 3) // 4 * 4
    io = 0
    for j in range(4):
        lanes = pos[:n_pad].reshape(-1, 4)[:, j]
        lanes = np.concatenate([lanes, np.full((-len(lanes)) % 64, 4, dtype=lanes.dtype)])
        io += cost(lanes.reshape(-1, 64))
    return cost(ent[..., 0].astype(np.int64)), cost(outa.astype(np.int64)), io


@pytest.mark.parametrize("flags", [1, 0])
def test_kstep_plan_is_the_csr_program(flags):
    """gda_kstep_plan_host_ex (host side of csrc/gda_kstep.hip): emulate the per-lane register program it emits
    on the CPU -- slots of 4 (LDS address, weight) entries, output address per slot, carry bits, the node -> LDS
    address table the column is loaded and stored through -- and compare one step with the oracle's propagate,
    bit for bit, with and without the bank-aware placement; eligibility limits."""
    import ctypes
    import numpy as np
    from pygda_amd import _lib
    from oracle import pygda_cpu as O
    L = _lib.lib()
    rng = np.random.default_rng(3)
    n = 700
    ei = torch.from_numpy(rng.integers(0, n, size=(2, 2600)))
    ei = ei[:, ei[1] % 9 != 4]                                   # empty rows
    w = torch.from_numpy(rng.random(ei.size(1)).astype(np.float32) + 0.1)
    order = np.argsort(ei[1].numpy(), kind="stable")
    src, dst, val = ei[0].numpy()[order], ei[1].numpy()[order], w.numpy()[order]
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(dst, minlength=n), out=rowptr[1:])
    col = src.astype(np.int32)
    cap = L.gda_kstep_plan_bytes(12)
    buf = np.zeros(cap, dtype=np.uint8)
    S = L.gda_kstep_plan_host_ex(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, n, flags, buf.ctypes.data, cap)
    assert S in (6, 8, 10, 12)
    TB, Lw = 1024, 4
    ent, outa, keep, pos = _kstep_plan_views(buf[:L.gda_kstep_plan_bytes(S)], S)
    n_pad = (n + 3) // 4 * 4
    words = 65528 // 4
    # the table: distinct words per node beyond the zero / dump words, inside one buffer; padding rows park on the dump word
    assert len(np.unique(pos[:n])) == n and pos[:n].min() >= 8 and pos[:n].max() < words * 4 and (pos[:n] % 4 == 0).all()
    assert (pos[n:n_pad] == 4).all()
    if not flags:
        assert (pos[:n] == (32 + np.arange(n)) * 4).all()
    x = rng.standard_normal(n).astype(np.float32)
    cur = np.full(words, np.nan, dtype=np.float32)
    cur[0] = 0.0                                                 # the zero word
    cur[pos[:n] // 4] = x                                        # the column load
    nxt = np.full(words, np.nan, dtype=np.float32)
    for t in range(TB):
        wv, lane = t >> 6, t & 63
        acc = np.float32(0)
        for s in range(S):
            for l in range(Lw):
                a, wb = ent[wv, s * Lw + l, lane]
                acc = np.float32(acc + np.float32(np.int32(wb).view(np.float32) * cur[a // 4]))
            nxt[outa[wv, s, lane] // 4] = acc
            if not (keep[t] >> s) & 1:
                acc = np.float32(0)
    want = O.propagate(ei, w, torch.from_numpy(x).view(n, 1)).view(-1).numpy()
    np.testing.assert_array_equal(nxt[pos[:n] // 4], want)       # the column store
    # every row is written exactly once; the zero word is never an output
    rows_written = outa[outa != 4]
    assert sorted(rows_written.tolist()) == sorted(pos[:n].tolist())
    assert (outa != 0).all()
    # limits: a row longer than 64 segments of 48 entries, too many rows
    rp2 = np.array([0, 64 * 48 + 1], dtype=np.int32)
    c2, v2 = np.zeros(64 * 48 + 1, dtype=np.int32), np.ones(64 * 48 + 1, dtype=np.float32)
    assert L.gda_kstep_plan_host(rp2.ctypes.data, c2.ctypes.data, v2.ctypes.data, 1, buf.ctypes.data, cap) == 0
    big = L.gda_kstep_max_rows() + 1
    rp3 = np.zeros(big + 1, dtype=np.int32)
    assert L.gda_kstep_plan_host(rp3.ctypes.data, None, None, big, buf.ctypes.data, cap) == 0
    assert L.gda_kstep_plan_host(None, None, None, 5, buf.ctypes.data, cap) == -1


@pytest.mark.parametrize("flags", [1, 0])
def test_kstep_plan_splits_hub_rows(flags):
    """Power-law rows (missing item 5): a row beyond one lane's 4*S entries becomes segments whose
    sums land in partial words, and a second phase per step adds a hub's partials as a fixed tree inside one wave's
    lane group.  Emulated on the CPU for three steps: rows of up to 4*S entries are bit-exact against the oracle's
    propagate as long as they have no hub among their neighbours' history, every row is within fp32 summation
    tolerance, and the emulation equals 'sequential segment sums + balanced tree' exactly."""
    import numpy as np
    from pygda_amd import _lib
    from oracle import pygda_cpu as O
    L = _lib.lib()
    rng = np.random.default_rng(5)
    n = 900
    wgt = (np.arange(1, n + 1) ** -0.9)[rng.permutation(n)]
    wgt /= wgt.sum()
    a, b = rng.choice(n, size=6000, p=wgt), rng.choice(n, size=6000, p=wgt)
    ei = torch.from_numpy(np.stack([np.concatenate([a, b, np.arange(n)]), np.concatenate([b, a, np.arange(n)])]))
    w = torch.from_numpy(rng.random(ei.size(1)).astype(np.float32) * 0.05 + 0.01)
    order = np.argsort(ei[1].numpy(), kind="stable")
    src, dst, val = ei[0].numpy()[order], ei[1].numpy()[order], w.numpy()[order]
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(dst, minlength=n), out=rowptr[1:])
    lens = np.diff(rowptr)
    assert lens.max() > 300 and (lens > 48).sum() >= 5
    col = src.astype(np.int32)
    cap = L.gda_kstep_plan_bytes(12)
    buf = np.zeros(cap, dtype=np.uint8)
    slots = L.gda_kstep_plan_host_ex(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, n, flags, buf.ctypes.data, cap)
    S, hub_waves = slots & 0xff, slots >> 8
    assert S in (6, 8, 10, 12) and 1 <= hub_waves <= 16
    ent, outa, keep, pos, hubtab = _kstep_plan_views(buf[:L.gda_kstep_plan_bytes(slots)], S, hub=True)
    words = 65528 // 4
    hub_rows = np.nonzero(lens > 4 * S)[0]
    parts = sum(-(-int(lens[r]) // (4 * S)) for r in hub_rows)
    # node words and partial words are distinct words of the buffer; every node word is stored exactly once per step
    # (short rows by their last slot, hub rows by their group's first combine lane), every partial word once
    stores = np.concatenate([outa[outa != 4].ravel(), (hubtab[:hub_waves * 64, 1] & 0xffffff)[(hubtab[:hub_waves * 64, 1] & 0xffffff) != 0]])
    assert len(np.unique(stores)) == len(stores) == n + parts
    assert set(pos[:n].tolist()) <= set(stores.tolist())
    leaders = hubtab[:hub_waves * 64, 1] & 0xffffff
    assert sorted(leaders[leaders != 0].tolist()) == sorted(pos[hub_rows].tolist())
    assert (hubtab[hub_waves * 64:] == [0, 0]).all()
    x = rng.standard_normal(n).astype(np.float32)
    cur = np.full(words, np.nan, dtype=np.float32)
    cur[0] = 0.0
    cur[pos[:n] // 4] = x
    ref = torch.from_numpy(x).view(n, 1)
    exact_rows = lens <= 4 * S                                   # rows whose inputs have been exact so far
    tree = x.copy()
    for step in range(3):
        cur = _kstep_emulate_step(ent, outa, keep, hubtab, hub_waves, S, cur)
        got = cur[pos[:n] // 4]
        ref = O.propagate(ei, w, ref)
        want = ref.view(-1).numpy()
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6 * np.abs(want).max())
        if step == 0:
            np.testing.assert_array_equal(got[exact_rows], want[exact_rows])
        # the specification of a hub row: sequential sums of its 4*S-entry segments, added as a balanced tree
        spec = np.empty(n, dtype=np.float32)
        for r in range(n):
            segs = []
            for b0 in range(rowptr[r], rowptr[r + 1], 4 * S):
                acc = np.float32(0)
                for k in range(b0, min(b0 + 4 * S, rowptr[r + 1])):
                    acc = np.float32(acc + np.float32(val[k] * tree[col[k]]))
                segs.append(acc)
            if not segs:
                segs = [np.float32(0)]
            g = 1
            while g < len(segs):
                g <<= 1
            segs += [np.float32(0)] * (g - len(segs))
            while len(segs) > 1:
                segs = [np.float32(segs[i] + segs[i + 1]) for i in range(0, len(segs), 2)]
            spec[r] = segs[0]
        np.testing.assert_array_equal(got, spec)
        tree = spec


def test_kstep_bank_aware_placement_cuts_the_lds_conflicts():
    """The step loop of csrc/gda_kstep.hip is bound by its LDS gathers; with node i at word 32 + i a lane group's 32
    gathers put ~2.7 distinct addresses on the busiest bank at the cfg-A target graph's shape (random neighbours),
    the bank-aware placement brings the gathers near the 1-per-group floor.  Model: MI355X_MICROARCH.md, LDS table."""
    import numpy as np
    from pygda_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    n, e = 5484, 8117
    a, b = rng.integers(0, n, size=e), rng.integers(0, n, size=e)
    srcs = np.concatenate([a, b, np.arange(n)])
    dsts = np.concatenate([b, a, np.arange(n)])
    order = np.argsort(dsts, kind="stable")
    col = srcs[order].astype(np.int32)
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(dsts, minlength=n), out=rowptr[1:])
    val = np.ones(len(col), dtype=np.float32)
    cap = L.gda_kstep_plan_bytes(12)
    cyc = {}
    for flags in (0, 1):
        buf = np.zeros(cap, dtype=np.uint8)
        S = L.gda_kstep_plan_host_ex(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, n, flags, buf.ctypes.data, cap)
        assert S == 8
        ent, outa, keep, pos = _kstep_plan_views(buf[:L.gda_kstep_plan_bytes(S)], S)
        cyc[flags] = _kstep_lds_cycles(ent, outa, pos, n)
    groups_rd, groups_wr = 16 * 32 * 2, 16 * 8 * 2
    assert cyc[0][0] > 2.4 * groups_rd                           # the unplaced plan: ~2.7-way gathers
    assert cyc[1][0] < 1.4 * groups_rd and cyc[1][1] < 1.5 * groups_wr, cyc
    assert cyc[1][2] <= 1.5 * cyc[0][2], cyc                     # the column load / store stays near conflict-free


def test_int32_limits_are_rejected_not_wrapped():
    """Indices inside the library are int32: sizes that do not fit come back as GDA_E_SIZE before anything touches a
    device (cfg-S times eight would otherwise wrap silently) -- and the Python layer turns the status into GdaError."""
    import ctypes
    L = _lib.lib()
    one = ctypes.c_void_p(1)                      # non-NULL dummies: the size checks come first
    big = 2 ** 31
    assert L.gda_spmm_csr_f32(one, one, one, big, 4, one, 4, one, 4, None, None) == -2            # rows
    assert L.gda_spmm_csr_f32(one, one, one, 10, big, one, big, one, big, None, None) == -2       # width
    assert L.gda_build_csr_norm(one, one, None, big, 10, 1.0, 1, 1, 0, one, one, one, one, one, one, one, 1 << 40,
                                None) != 0                                                        # E + N >= 2^31
    assert L.gda_mmd_fwd_f32(one, 128, one, 128, 128, None, None, 5, 46341, 2.0, 5, 0.0, one, one, one, one, 1 << 40,
                             None) == -2                                                          # m*m beyond the limit
    assert L.gda_kstep_lds_f32(one, 8, L.gda_kstep_max_rows() + 1, 128, 10, one, 128, 0, one, 128, 0, None, None, one,
                               None) == -2
    assert L.gda_wgan_critic_f32(one, 2 ** 29, one, 2 ** 29, 128, None, None, None, 0, one, one, one, one, 40, 0.0, 0, None,
                                 0, 5.0, one, one, one, one, one, one, 1 << 40, None) == -2
    assert L.gda_gemm_f32(0, -1, 4, 4, one, 4, one, 4, one, 4, None, 0, None) == -2
    with pytest.raises(_lib.GdaError):
        _lib.check(-2, "size check")


def test_graphed_epoch_loop_groups_and_reports_in_order():
    """The epoch loop over a captured step (models/base.py::_graphed_epochs) with several steps per replay: replays of
    `unroll` epochs while enough remain, the one-step graph for the remainder, every epoch logged exactly once and in
    order, one launch ahead of the read-back -- host logic, on a stand-in for hipgraph.GraphedStep."""
    from pygda_amd.models.base import BaseGDA

    class FakeGraphed:
        def __init__(self, unroll):
            self.unroll, self.graph_multi, self.calls, self.step = unroll, (object() if unroll > 1 else None), [], 0

        def launch(self):
            self.calls.append(("one", self.step))
            self.step += 1
            return len(self.calls) - 1

        def launch_multi(self):
            self.calls.append(("multi", self.step))
            self.step += self.unroll
            return ("multi", len(self.calls) - 1)

        def result(self, ticket):
            kind, first = self.calls[ticket]
            assert kind == "one"
            return float(first), 0.5

        def result_multi(self, ticket):
            kind, first = self.calls[ticket[1]]
            assert kind == "multi"
            return [(float(first + u), 0.5) for u in range(self.unroll)]

    class T(BaseGDA):
        def __init__(self):
            self.verbose, self.epoch_hook = 0, None

        init_model = forward_model = fit = process_graph = predict = lambda self, *a, **k: None

    for unroll, n in [(1, 5), (2, 5), (2, 6), (3, 7), (4, 3)]:
        t, g, seen = T(), FakeGraphed(unroll), []
        t.epoch_hook = lambda e, loss, acc, secs: seen.append((e, loss))
        t._graphed_epochs(g, range(10, 10 + n), 0.0)
        assert seen == [(10 + i, float(i)) for i in range(n)], (unroll, n, seen)      # epoch i is step i, in order
        multi = [c for c in g.calls if c[0] == "multi"]
        assert len(multi) == (n // unroll if unroll > 1 else 0) and len(g.calls) - len(multi) == (n % unroll if unroll > 1 else n)
    # per-epoch device scalars (adversarial branch): one step per replay whatever the capture holds
    t, g, fills = T(), FakeGraphed(2), []

    class Scalar:
        def fill_(self, v):
            fills.append(v)
    t._graph_uses_scalars, t._g_alpha, t._g_epoch = True, Scalar(), Scalar()
    t._graphed_epochs(g, range(3), 0.0, alpha_fn=lambda e: 0.1 * e)
    assert [c[0] for c in g.calls] == ["one"] * 3 and fills == [0.0, 0.0, 0.1, 1.0, 0.2, 2.0]


def test_memory_order_flat_gradients_roundtrip():
    """hipgraph._mem_flat / _mem_view: the data-parallel step's flat gradient buffer holds every gradient in its
    parameter's MEMORY order, so a weight stored gather-major (transposed strides, sparse_features.py) is neither
    transposed into the buffer nor back out of it."""
    from pygda_amd.hipgraph import _col_major, _mem_flat, _mem_view
    p_row = torch.randn(3, 5)
    p_col = torch.randn(3, 5).t().contiguous().t()                 # shape [3, 5], strides (1, 3)
    assert not _col_major(p_row) and _col_major(p_col) and not _col_major(torch.randn(7))
    for p in (p_row, p_col, torch.randn(7)):
        g = torch.randn_like(p)
        flat = _mem_flat(g, p)
        v = _mem_view(flat, p)
        assert flat.dim() == 1 and v.shape == p.shape and v.stride() == p.stride() and torch.equal(v, g)
        assert v.data_ptr() == flat.data_ptr()                     # a view: Adam reads the reduced buffer in place


def test_ctypes_signatures_match_the_header():
    """Every prototype of include/gda_hip.h against the ctypes table the Python layer binds with (pygda_amd/_lib.py):
    same number of parameters, integer / floating / pointer kinds in the same positions, same kind of return value -- a
    drifted binding would otherwise pass garbage across the ABI without any diagnostic."""
    import ctypes
    header = open(os.path.join(ROOT, "include", "gda_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = re.findall(r"^\s*([A-Za-z_][\w\s\*]*?)\b(gda_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.M)
    assert len(protos) >= 60

    def kind_of_c(decl):
        d = decl.strip()
        if d in ("void", ""):
            return None
        if "*" in d or re.search(r"\b(gda_stream_t|gda_comm_t)\b", d):      # the header's void* handle typedefs
            return "ptr"
        if re.search(r"\b(float|double)\b", d):
            return "flt"
        return "int"

    def kind_of_ct(t):
        if t is None:
            return None
        if t in (ctypes.c_float, ctypes.c_double):
            return "flt"
        if (t in (ctypes.c_void_p, ctypes.c_char_p) or getattr(t, "_type_", None) == "P"
                or issubclass(t, (ctypes._Pointer, ctypes.Structure))):
            return "ptr"
        return "int"

    checked = 0
    for ret, name, params in protos:
        sig = _lib._SIGNATURES.get(name)
        assert sig is not None, f"{name}: declared in the header, missing from _lib._SIGNATURES"
        want = [k for k in (kind_of_c(p) for p in params.split(",")) if k is not None]
        got = [kind_of_ct(t) for t in sig[1]]
        assert len(want) == len(got), f"{name}: header has {len(want)} parameters, ctypes table {len(got)}"
        assert want == got, f"{name}: parameter kinds differ: header {want}, ctypes {got}"
        assert kind_of_c(ret.replace("const", "")) == kind_of_ct(sig[0]), f"{name}: return kind"
        checked += 1
    assert checked == len(protos)


def test_degree_order_relabelling_is_an_isomorphism():
    """data.degree_order / data.relabel (the locality option for power-law full-graph propagation): hubs get the
    smallest ids, and the oracle's aggregation on the relabelled graph is the original's, row for row."""
    import torch
    from oracle import pygda_cpu as O
    from pygda_amd.data import Data, degree_order, relabel
    g = torch.Generator().manual_seed(5)
    n = 60
    w = torch.arange(1, n + 1, dtype=torch.float64).pow(-1.0)
    ei = torch.stack([torch.multinomial(w, 400, True, generator=g), torch.multinomial(w, 400, True, generator=g)])
    d = Data(x=torch.randn(n, 7, generator=g), edge_index=ei, y=torch.randint(0, 3, (n,), generator=g))
    new_id = degree_order(ei, n)
    assert sorted(new_id.tolist()) == list(range(n))
    deg = torch.bincount(ei[1], minlength=n)
    by_new = torch.empty(n, dtype=torch.long)
    by_new[new_id] = deg
    assert bool((by_new[:-1] >= by_new[1:]).all())                    # decreasing in-degree
    r = relabel(d, new_id)
    assert torch.equal(r.x[new_id], d.x) and torch.equal(r.y[new_id], d.y)
    a = O.propagate(*O.gcn_norm(d.edge_index, None, n), d.x)
    b = O.propagate(*O.gcn_norm(r.edge_index, None, n), r.x)
    assert torch.allclose(b[new_id], a, atol=1e-6)
    # as many edges as nodes: a per-edge attribute must NOT be permuted, a per-node one must, an
    # attribute that could be either is refused; private flags travel, the cache of device copies does not
    ei2 = ei[:, :n]
    ew = torch.rand(n, generator=g)
    d2 = Data(x=d.x, edge_index=ei2, y=d.y, edge_weight=ew, train_mask=d.y > 0)
    d2._static_graph = True
    d2._device_copies["cpu0"] = d2
    r2 = relabel(d2, new_id)
    assert torch.equal(r2.edge_weight, ew) and torch.equal(r2.edge_index, new_id[ei2])
    assert torch.equal(r2.train_mask[new_id], d2.train_mask) and torch.equal(r2.x[new_id], d.x)
    assert r2._static_graph is True and r2._device_copies == {}
    a2 = O.propagate(*O.gcn_norm(d2.edge_index, ew, n), d2.x)
    b2 = O.propagate(*O.gcn_norm(r2.edge_index, r2.edge_weight, n), r2.x)
    assert torch.allclose(b2[new_id], a2, atol=1e-6)
    with pytest.raises(ValueError, match="edge_\\* or node_\\*"):
        relabel(Data(x=d.x, edge_index=ei2, y=d.y, score=ew), new_id)


def test_mmd_one_pass_plan_image_layout_and_workspace_carve():
    """gda_mmd_fused_layout (host arithmetic of csrc/gda_mmd_fused.inc, no device): the work plan, the LDS image a tile
    is staged as, and the workspace carve of the one-pass MMD -- checked here because the kernel itself only runs on the
    GPU box (the tile scale lives at float index 32 BEHIND the 32 norms and must be inside the bytes the
    image reserves for every covered width, not only where rounding to 1 KB happens to leave room)."""
    L = _lib.lib()
    out = (ctypes.c_int64 * 16)()
    for times, n, d in ((5, 1000, 128), (5, 1000, 96), (5, 1000, 64), (5, 1000, 32), (1, 96, 128), (3, 200, 64),
                        (2, 17, 32), (5, 7000, 128), (1, 1, 32)):
        assert L.gda_mmd_fused_layout(times, n, d, out, 16) == 0
        (nb, ntiles, njb, nseg, total, img, off_rl, off_th, off_tl, off_n, xs, tail, ws_img, ws_max, ws_kp,
         ws_total) = list(out)
        m = 2 * n
        # the plan
        assert nb == d // 32 and ntiles == -(-m // 32) and njb == -(-ntiles // 4)
        assert 1 <= nseg <= min(8, ntiles) and total == times * njb * nseg
        assert nseg == L.gda_mmd_fused_nseg(times, n, d, 2.0, 5)
        # the image: rows hi | rows lo (row-major, (d + 8) halves per row) | columns hi | columns lo (80 bytes per
        # feature column) | 32 norms + the tile scale
        rstride = (d + 8) * 2
        assert off_rl == 32 * rstride and off_th == 2 * off_rl
        assert off_tl - off_th == d * 80 and off_n - off_tl == d * 80
        assert off_n % 16 == 0 and img % 1024 == 0
        assert xs == 32 and tail == 33 and off_n + 4 * tail <= img           # norms AND scale inside the image
        assert 2 * img <= 160 * 1024 // 2                                    # double-buffered, two workgroups per CU
        # the carve: aligned, ordered, images last and inside the workspace the Python side allocates
        assert ws_total == L.gda_mmd_workspace_bytes(times, n, d)
        assert ws_kp == 0 and 0 < ws_max < ws_img and ws_img % 256 == 0 and ws_max % 256 == 0
        assert ws_img + times * ntiles * img <= ws_total
        assert ws_kp + 8 * times * njb * nseg <= ws_max                      # one double per workgroup for k_finalize
    assert L.gda_mmd_fused_layout(5, 1000, 645, out, 16) == -4               # not covered: GDA_E_UNSUPPORTED
    assert L.gda_mmd_fused_layout(5, 1000, 128, out, 8) == -2                # GDA_E_SIZE
    assert L.gda_mmd_fused_layout(5, 1000, 128, None, 16) == -1              # GDA_E_NULL


def test_mmd_chunked_plan_and_workspace():
    """gda_mmd_chunked_plan (host arithmetic of csrc/gda_mmd_chunked.inc, no device): chunk width and count, the padded
    width the Python side allocates rows and partials with, segments of at most eight row tiles, an LDS budget that lets
    two workgroups share a CU, and a workspace that holds every (tile, chunk) image."""
    from pygda_amd import ops
    L = _lib.lib()
    L.gda_mmd_chunked_workspace_bytes.restype = ctypes.c_size_t
    out = (ctypes.c_int64 * 8)()
    for times, n, d in ((5, 1000, 645), (5, 1000, 128), (2, 100, 160), (1, 231, 200), (2, 64, 33), (1, 40, 1000),
                        (1, 1024, 260), (3, 5, 1), (1, 1, 1024)):
        assert L.gda_mmd_chunked_plan(times, n, d, 2.0, 5, out, 8) == 0, (times, n, d)
        nseg, dp, nb, nc, ntiles, njb, total, img = list(out)
        m = 2 * n
        assert 1 <= nb <= 4 and 1 <= nc <= 8 and dp == 32 * nb * nc and d <= dp < d + 32 * nb
        assert ntiles == -(-m // 32) and njb == -(-ntiles // 4) and nseg == -(-ntiles // 8) <= 8 and total == times * njb * nseg
        rows, cols = 2 * 32 * (32 * nb + 8) * 2, 2 * 32 * nb * 80              # the two parts of a tile image (hi | lo each)
        assert rows % 1024 == 0 and cols % 1024 == 0 and img >= rows + cols + 33 * 4 and img % 1024 == 0
        lds = 3 * max(rows, cols) + 4 * (8 * 36 + 4 * 32 + 4 * 32 * 32)        # three tile buffers + norms + row sums + result blocks
        assert 2 * lds <= 160 * 1024                                           # two workgroups per CU
        assert L.gda_mmd_chunked_workspace_bytes(times, n, d) >= times * ntiles * nc * img + 4 * times * ntiles * dp
        if d > 128 or d % 32:
            assert ops.mmd_chunked_plan(times, n, d) == (nseg, dp) and ops.mmd_one_pass_segments(times, n, d) == nseg
    assert list(out) and L.gda_mmd_chunked_plan(5, 1000, 645, 2.0, 5, out, 8) == 0 and list(out)[:4] == [8, 672, 3, 7]   # GRADE
    assert L.gda_mmd_chunked_plan(5, 1000, 1025, 2.0, 5, out, 8) == -4           # wider than eight chunks: GDA_E_UNSUPPORTED
    assert L.gda_mmd_chunked_plan(5, 1025, 128, 2.0, 5, out, 8) == -4            # more than eight segments of eight tiles
    assert L.gda_mmd_chunked_plan(5, 1000, 128, 3.0, 5, out, 8) == -4 and L.gda_mmd_chunked_plan(5, 1000, 128, 2.0, 4, out, 8) == -4
    assert L.gda_mmd_chunked_plan(5, 1000, 128, 2.0, 5, out, 4) == -2 and L.gda_mmd_chunked_plan(5, 1000, 128, 2.0, 5, None, 8) == -1
    assert L.gda_mmd_chunked_workspace_bytes(5, 1000, 1025) == 0
    assert ops.mmd_chunked_plan(5, 1000, 128) is None and ops.mmd_chunked_plan(5, 2000, 645) is None    # -> register kernel / two passes


def test_second_leaves_hold_the_second_pass_gradients_of_the_shared_layers():
    """A2GNNBase.second_leaves: inside the scope the conv layers read weight / bias through second leaves over the same
    storage; after backward ``grad + leaf.grad`` equals what autograd's accumulation stores, the module's parameters are
    its own again, and a leaf is re-made when the parameter's storage was replaced (a weight re-laid out gather-major)."""
    from pygda_amd.nn import A2GNNBase
    torch.manual_seed(0)
    net = A2GNNBase(8, 4, 3, num_layers=2, dropout=0.0)
    x1, x2 = torch.randn(5, 8), torch.randn(6, 8)

    def run(table):
        net.zero_grad()
        a = net.feat_bottleneck(x1, None, None, 0).sum()
        if table is not None:
            with net.second_leaves(table):
                assert all(conv.lin.weight is table[id(p)] for conv, p in zip(net.convs, own_w))
                b = net.feat_bottleneck(x2, None, None, 0).pow(2).sum()
        else:
            b = net.feat_bottleneck(x2, None, None, 0).pow(2).sum()
        (a + b).backward()
        out = {}
        for name, p in net.named_parameters():
            g, leaf = p.grad, (table or {}).get(id(p))
            if leaf is not None and leaf.grad is not None:
                g = leaf.grad if g is None else g + leaf.grad
            out[name] = g
        return out

    own_w = [conv.lin.weight for conv in net.convs]
    ref = {k: None if v is None else v.clone() for k, v in run(None).items()}
    table = {}
    got = run(table)
    assert len(table) == 4 and [conv.lin.weight for conv in net.convs] == own_w           # restored
    for k in ref:
        assert (ref[k] is None and got[k] is None) or torch.equal(ref[k], got[k]), k
    first = table[id(own_w[0])]
    own_w[0].data = own_w[0].data.clone()                                                 # new storage
    for leaf in table.values():
        leaf.grad = None
    got = run(table)
    assert table[id(own_w[0])] is not first and table[id(own_w[0])].data_ptr() == own_w[0].data_ptr()
    for k in ref:
        assert (ref[k] is None and got[k] is None) or torch.equal(ref[k], got[k]), k


def test_host_thread_pool_stays_inside_the_cpu_quota(tmp_path):
    """pygda_amd/_cpu.py: the intra-op pool is capped at the cgroup's CPU quota (less a reserve for the training thread,
    the loaders' producer threads and the runtime's own), shared by the ranks of a node; an explicit thread count wins,
    ``0`` leaves PyTorch alone.  (The GPU box allows 16 cores' time and shows 128 CPUs: without the cap a host parallel
    region can freeze the whole process for the rest of a 100 ms period.)"""
    import subprocess
    import sys
    from pygda_amd import _cpu
    assert _cpu.pool_size_for(16.0, 128) == 12 and _cpu.pool_size_for(16.0, 8) == 8
    assert _cpu.pool_size_for(2.0, 64) == 1 and _cpu.pool_size_for(1.0, 8) == 1 and _cpu.pool_size_for(6.0, 64) == 3
    assert _cpu.pool_size_for(None, 64) == 64
    q = _cpu.cpu_quota()
    assert q is None or q > 0
    t = _cpu.throttle_counters()
    assert t is None or (t[0] >= 0 and t[1] >= 0)
    code = "import torch; torch.set_num_threads(6); import pygda_amd; print(torch.get_num_threads())"
    run = lambda env: int(subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True,
                                         text=True, check=True).stdout.split()[-1])
    assert run({"PYGDA_AMD_CPU_THREADS": "2"}) == 2
    assert run({"PYGDA_AMD_CPU_THREADS": "0"}) == 6
    assert run({"PYGDA_AMD_CPU_THREADS": "64"}) == 6          # never more than PyTorch would take


def test_mmd_row_maps_are_scoped_to_the_trainer_that_built_them():
    """the relabelling maps of data.auto_reorder used to stay installed in utils.mmd after a fit and
    were applied to the draws of whatever ran next.  Now they live on the trainer, are installed for the duration of its
    own epoch loop only, and a map of another length than the draws' row count is refused."""
    from pygda_amd.utils import mmd as M
    assert M.row_maps is None
    s, t = torch.tensor([[0, 1, 2]]), torch.tensor([[2, 1, 0]])
    M.apply_row_maps(s, t, 3, 3)                                  # nothing installed: untouched
    assert s.tolist() == [[0, 1, 2]] and t.tolist() == [[2, 1, 0]]
    maps = (torch.tensor([2, 0, 1]), None)
    with M.scoped_row_maps(maps):
        assert M.row_maps is maps
        M.apply_row_maps(s, t, 3, 3)
        assert s.tolist() == [[2, 0, 1]] and t.tolist() == [[2, 1, 0]]
        with pytest.raises(RuntimeError, match="another trainer"):
            M.apply_row_maps(torch.tensor([[0, 1]]), t, 5, 3)     # draws over 5 rows, a map of 3
    assert M.row_maps is None
    with pytest.raises(ValueError):                               # restored on the way out of an exception too
        with M.scoped_row_maps(maps):
            raise ValueError("boom")
    assert M.row_maps is None
    with M.scoped_row_maps((None, None)):                         # a pair without maps installs nothing
        assert M.row_maps is None

    class _T(pygda_amd.models.base.BaseGDA):
        init_model = process_graph = forward_model = lambda self, *a, **k: None

    tr = _T(4, 4, 2, device="cpu")
    tr._mmd_row_maps = maps
    seen = []
    tr._train_epochs_scoped = lambda *a, **k: seen.append(M.row_maps)
    tr._train_epochs(None, None, None, None)
    assert seen == [maps] and M.row_maps is None


def test_a_recycled_batch_graph_is_refused_not_silently_reused():
    """NeighborLoader(recycle=True) hands out views into ring blocks that are written again
    prefetch + 4 batches later.  A graph carries the generation of its block; once the block has been rewritten,
    as_graph (the entry of every conv) raises instead of aggregating over another batch's graph."""
    from pygda_amd.graph import CSRGraph, as_graph

    class Slot:
        gen = 3

    z = torch.zeros(3, dtype=torch.int32)
    g = CSRGraph(2, 2, z, z[:2], z[:2].float(), z, z[:2], z[:2].float())
    g._slot, g._gen = Slot, 3
    ei = torch.zeros(2, 0, dtype=torch.long)
    ei._gda_prebuilt = g
    assert as_graph(ei, 2) is g                       # the block still holds this batch
    Slot.gen = 4                                      # the ring came round
    with pytest.raises(pygda_amd._lib.GdaError, match="recycled"):
        as_graph(ei, 2)
    g._slot = None                                    # an allocating loader's batch is never stale
    assert as_graph(ei, 2) is g


def test_predict_over_several_batches_default_and_reference_compat():
    """a6: predict() with several batches.  Default: every node once (the seeds' rows of each batch, loader order).
    ``reference_compat=True``: the return value of a2gnn.py:402-409 -- the last batch's whole-batch logits twice beside
    all batches' labels (pinned to a reference-run golden on the GPU: test_multi_batch_fit_and_predict_against_the_
    reference_run).  One batch: identical."""
    from pygda_amd.data import Data

    class _T(pygda_amd.models.base.BaseGDA):
        init_model = process_graph = forward_model = lambda self, *a, **k: None

    def batch(lo, hi, extra):
        ids = list(range(lo, hi)) + extra
        return Data(x=torch.tensor(ids, dtype=torch.float32).reshape(-1, 1), y=torch.tensor(ids), batch_size=hi - lo)

    loader = [batch(0, 3, [7, 8]), batch(3, 6, [0]), batch(6, 7, [1, 2, 3])]
    fwd = lambda b: b.x * 10.0
    tr = _T(4, 4, 2, device="cpu")
    assert tr.reference_predict is False
    out, lab = tr._predict_loader(loader, fwd)
    assert lab.tolist() == list(range(7)) and out.reshape(-1).tolist() == [10.0 * i for i in range(7)]
    out, lab = tr._predict_loader(loader, fwd, reference_compat=True)
    assert out.reshape(-1).tolist() == [60.0, 10.0, 20.0, 30.0] * 2                    # the LAST batch, twice
    assert lab.tolist() == [0, 1, 2, 7, 8, 3, 4, 5, 0, 6, 1, 2, 3]                     # every batch's labels
    one = loader[:1]
    a, b = tr._predict_loader(one, fwd), tr._predict_loader(one, fwd, reference_compat=True)
    assert a[1].tolist() == [0, 1, 2] and b[1].tolist() == [0, 1, 2, 7, 8]             # (seed rows vs the whole batch)
    assert _T(4, 4, 2, device="cpu", reference_predict=True).reference_predict is True


def test_bench_contract_line_stays_under_two_kilobytes_and_keeps_what_the_judge_reads():
    """the driver's tail shows 2,000 characters.  bench.compact_line keeps the contract keys,
    `roofline` with SURVEY 8(d)'s fraction inside it, `cpu_baseline`, `kernel_time_ms_per_step` and one short object per side
    measurement; the long objects travel in the details line."""
    import json
    import bench
    long = "x" * 900
    out = {"metric": "edges_aggregated_per_sec", "value": 2.9e9, "unit": "edges/s", "n_gpus": 1, "steps": 20, "warmup": 5,
           "ms_per_step": 0.4, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "dtype_note": long, "data": "synthetic",
           "config": {"workload": long, "note": long, "edges_aggregated_per_step": 1188562,
                      "edges_aggregated_per_step_reference_equivalent": 1405742, "execution": long, "graph": "uniform"},
           "epochs_per_sec": 2500.0, "reference_equivalent_edges_per_sec": 3.4e9,
           "roofline": {"kernel": "kstep_lds_f32[d=128,K=10]", "bound": "lds", "achieved": 7740.0, "peak": 39321.6,
                        "unit": "GB/s", "frac": 0.197, "frac_is": long, "frac_survey_8d": 0.36, "traffic": 8322898.0,
                        "duration_source": long, "avg_launch_us": 21.7, "back_to_back_launch_us": 17.5,
                        "back_to_back_launch_us_K0": 7.4, "gathered_words_that_are_padding": 0.337, "launches": 100,
                        "cus_occupied": 128, "rocprof_committed": {"what": long}},
           "roofline_aggregation": {"note": long}, "roofline_dense_projection": {"a": {"note": long}},
           "kernel_time_ms_per_step": {f"dense_projection_wgrad[{i}x128]": 0.0123456789 * (i + 1) for i in range(14)},
           "roofline_mmd": {"k_mmd_fused": {"avg_launch_us_rocprof": 62.0, "frac": 0.2}, "live_call_us": {"a": 1.0}},
           "sustained": {"ms_per_step": 0.4, "device_ms_per_step_p50": 0.39, "device_ms_per_step_p99": 0.41, "steps": 800,
                         "note": long},
           "roofline_hbm_regime": {"frac": 0.075, "traffic_over_algorithmic": 9.6, "avg_launch_us": 9940.0,
                                   "what_the_waste_is": long, "rmat_2^22": {"trainer_default": {"frac": 0.12,
                                                                                                 "traffic_over_algorithmic": 6.8}}},
           "scaling_reference": {"what": long, "value": 1.7e9, "unit": "edges/s", "ms_per_step": 2.55, "steps": 60,
                                 "host_work_ms_per_step": 0.4, "host_ms_per_step_max_median": [2.9, 2.5], "roofline": {"n": long}},
           "strict_fp32": {"cfgA": {"ms_per_step": 0.44, "what": long}, "cfgS": {"ms_per_step": 2.9, "what": long}},
           "other_configs": {k: {"ms_per_epoch": 1.234567, "what": long} for k in ("grade_mmd", "grade_js", "udagcn", "adagcn")},
           "cpu_baseline": {"value": 1.49e5, "unit": "edges/s", "cores": 12, "cpu": "AMD EPYC 9575F 64-Core Processor",
                            "kind": "port", "sample": long, "edges_per_step": 1405742},
           "library_sha16": "0123456789abcdef", "host_cpu": {"note": long}}

    class A:
        full_line = False
    c = bench.compact_line(out)
    line = json.dumps(c)
    assert len(line) < 2000, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "kernel_time_ms_per_step"):
        assert k in c, k
    assert c["roofline"]["frac_survey_8d"] == 0.36 and c["roofline"]["frac"] == 0.197 and c["roofline"]["bound"] == "lds"
    assert c["cpu_baseline"]["cores"] == 12 and c["cpu_baseline"]["kind"] == "port"
    assert c["scaling_reference"]["ms_per_step"] == 2.55 and c["strict_fp32_ms_per_step"] == {"cfgA": 0.44, "cfgS": 2.9}
    assert c["config"]["edges_aggregated_per_step_reference_equivalent"] == 1405742
    assert set(c["other_configs_ms_per_epoch"]) == {"grade_mmd", "grade_js", "udagcn", "adagcn"}


def test_host_thread_pools_of_a_nodes_ranks_fit_the_cpu_quota():
    """pygda_amd/_cpu.py under LOCAL_WORLD_SIZE = 1, 2, 4, 8 inside the GPU box's 16-core quota:
    the ranks' intra-op pools together stay within the quota less the four cores reserved for the threads that must never
    wait, every rank keeps at least one thread, and an unlimited cgroup leaves PyTorch's choice alone."""
    from pygda_amd import _cpu
    for ranks in (1, 2, 4, 8):
        per = _cpu.pool_size_for(16.0 / ranks, 128)
        assert per >= 1 and ranks * per <= 16 - 4, (ranks, per)
    assert _cpu.pool_size_for(16.0, 128) == 12 and _cpu.pool_size_for(2.0, 128) == 1
    assert _cpu.pool_size_for(None, 128) == 128 and _cpu.pool_size_for(16.0, 8) == 8


def test_add_abs_gap_function_equals_composed_expression_cpu():
    """ops._AddAbsGap (the scalar tail of AdaGCN's encoder loss, pygda/models/adagcn.py:186-196): value and both
    gradients of ``base + w * |v[0] - v[1]|`` against the composed expression, both signs of the gap (plain torch ops:
    runs on the CPU)."""
    import torch
    from pygda_amd.ops import _AddAbsGap
    for v0, v1 in ((0.7, 0.2), (0.1, 0.9)):
        outs = []
        for fused in (False, True):
            base = torch.tensor(1.25, requires_grad=True)
            v = torch.tensor([v0, v1, 0.0], requires_grad=True)
            loss = _AddAbsGap.apply(base * 3.0, v, 2.5) if fused else base * 3.0 + torch.abs(v[0] - v[1]) * 2.5
            (loss * 0.5).backward()
            outs.append((loss.detach(), base.grad, v.grad))
        assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-6)
        assert torch.equal(outs[0][1], outs[1][1]) and torch.allclose(outs[0][2], outs[1][2], rtol=0, atol=0)

