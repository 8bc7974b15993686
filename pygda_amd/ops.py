"""torch.autograd wrappers over the C ABI (include/gda_hip.h).  PyTorch supplies device
memory, the stream and autograd plumbing; every numeric kernel named here is ours."""
import ctypes
import weakref

import os as _os

import torch

from . import _lib, profiler
from .graph import CSRGraph, build_csr


def _f32c(t, name):
    _lib.require_gpu_tensor(t, name)
    if t.dtype != torch.float32:
        raise _lib.GdaError(f"{name}: the aggregation path computes in fp32, got {t.dtype}")
    return t.contiguous()


# ------------------------------------------------- the activation's backward in its producer's epilogue --
# dropout(relu(.)) needs, going backwards, only its own OUTPUT y: gx = (y > 0) * g / (1 - p).  When the gradient g of y has
# exactly one producer and that producer is one of ours, it can apply the mask while it stores g -- the activation's own
# backward launch and one [rows, d] round trip through HBM disappear (cfg-S: ~0.2 ms of a 2.5 ms step).  Protocol: an
# activation that is willing hands its output a GradSink (buffer for the masked gradient, y, p); a sink-aware producer
# (`gemm(GEMM_NN, ..., sink=)`, the one-pass MMD's scatter) writes the MASKED gradient into `sink.buf`, sets `written` and
# returns that buffer; the activation's backward recognises its own buffer and passes it on untouched.  Anything else
# (another producer, autograd summing two producers into a fresh tensor) arrives as an ordinary gradient and takes the
# ordinary kernel.  ONLY sound when y has a single consumer: a second, sink-unaware consumer's gradient would be summed
# onto an already masked one and masked again.  So sinks are handed out only inside `grad_sinks()`, which the A2GNN
# trainer opens around its sampled step, where every activation output feeds exactly one op.
GRAD_SINKS = _os.environ.get("PYGDA_AMD_GRAD_SINKS", "1") == "1"
_sinks_on = False
sink_hits = 0            # activation backward launches that a producer's epilogue made unnecessary (tests read it)


class grad_sinks:
    """``with grad_sinks():`` -- activations created inside may hand out GradSinks (see above)."""

    def __enter__(self):
        global _sinks_on
        self.prev, _sinks_on = _sinks_on, GRAD_SINKS
        return self

    def __exit__(self, *exc):
        global _sinks_on
        _sinks_on = self.prev
        return False


class GradSink:
    __slots__ = ("buf", "y", "p", "written")

    def __init__(self, buf, y, p):
        # (an ALIAS of y, not y itself: y carries this sink as an attribute, and a tensor <-> sink reference cycle would
        # keep every eager step's activations alive until the cyclic collector runs -- which the training loops switch off)
        self.buf, self.y, self.p, self.written = buf, y.detach(), float(p), False

    def mine(self, g):
        """Is ``g`` the masked gradient a producer left in this sink?  (consumes the flag)"""
        global sink_hits
        hit = self.written and g is not None and g.data_ptr() == self.buf.data_ptr() and g.shape == self.buf.shape
        self.written = False
        sink_hits += bool(hit)
        return hit


def sink_of(t):
    return getattr(t, "_gda_grad_sink", None) if torch.is_tensor(t) else None


def _offer_sink(y, p, buf=None, wanted=True):
    """Attach a sink to the activation output ``y`` when sinks are on and the activation's input wants a gradient."""
    if _sinks_on and wanted and y.is_cuda and y.dim() == 2 and y.size(1) % 4 == 0 and y.is_contiguous():
        y._gda_grad_sink = GradSink(torch.empty_like(y) if buf is None else buf, y, p)
        return y._gda_grad_sink
    return None


def masked_dgrad(gy, weight, sink):
    """``sink.buf <- (sink.y > 0) * (gy @ weight) / (1 - p)`` in the product's epilogue when the kernel takes the shape,
    else product + activation backward as two launches; either way the sink is written."""
    gy, weight = _f32c(gy, "gy"), _f32c(weight, "weight")
    (M, K), (K2, N) = gy.shape, weight.shape
    L = _lib.lib()
    st = L.gda_gemm_nn_mask_f32(M, N, K, _lib.ptr(gy), K, _lib.ptr(weight), N, _lib.ptr(sink.buf), N, _lib.ptr(sink.y), N,
                                sink.p, _lib.stream()) if (M == sink.buf.size(0) and N == sink.buf.size(1)) else -1
    if st != 0:
        gx = gemm(GEMM_NN, gy, weight)
        _lib.check(L.gda_relu_dropout_bwd_f32(_lib.ptr(gx), _lib.ptr(sink.y), _lib.ptr(sink.buf), gx.numel(), sink.p,
                                              _lib.stream()), "gda_relu_dropout_bwd_f32")
    sink.written = True
    return sink.buf


# ------------------------------------------------------ source classification loss --
class _SoftmaxNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        x = _f32c(logits, "logits")
        _lib.require_gpu_tensor(labels, "labels", torch.int64)
        if x.dim() != 2 or labels.dim() != 1 or labels.numel() != x.size(0):
            raise ValueError(f"logits [N, C] and labels [N] expected, got {tuple(x.shape)} and {tuple(labels.shape)}")
        n, c = x.shape
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        stats = torch.empty(2, dtype=torch.float64, device=x.device)       # {loss, #(argmax == label)}: the epoch log
        L = _lib.lib()
        ws = _lib.workspace(L.gda_softmax_nll_workspace_bytes(), x.device, "ce")
        # a batch padded to a static capacity (pygda_amd/sampled_graph.py) tags its labels with the DEVICE count of its
        # real rows: the loss is their mean, the rows behind them get zero gradients
        ctx.nv = nv = getattr(labels, "_gda_valid_rows", None)
        _lib.check(L.gda_softmax_nll_fwd_nv_f32(_lib.ptr(x), c, _lib.ptr(labels.contiguous()), n, c, _lib.ptr(nv),
                                                _lib.ptr(loss), _lib.ptr(stats), _lib.ptr(ws), ws.numel(), _lib.stream()),
                   "gda_softmax_nll_fwd_nv_f32")
        global _ce_stats
        # weak references + version counters: the entry pins neither the [rows, C] logits nor the labels of a sampled
        # batch (the eager loop never looks it up), and a tensor that died -- whose storage may since have been handed
        # to another one -- or was written to can never match
        _ce_stats = (weakref.ref(logits), weakref.ref(labels), n, stats, logits._version, labels._version)
        ctx.save_for_backward(x, labels)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        x, labels = ctx.saved_tensors
        n, c = x.shape
        gx = torch.empty_like(x)
        gl = gl.reshape(1).to(torch.float32).contiguous()
        _lib.check(_lib.lib().gda_softmax_nll_bwd_nv_f32(_lib.ptr(x), c, _lib.ptr(labels.contiguous()), n, c,
                                                         _lib.ptr(ctx.nv), _lib.ptr(gl), _lib.ptr(gx), c, _lib.stream()),
                   "gda_softmax_nll_bwd_nv_f32")
        return gx, None


_ce_stats = None


def ce_stats_for(logits, labels):
    """``[loss, number of correct argmax predictions]`` (float64, device) left by the LAST softmax_nll call if it
    was made on exactly these logits and labels, else None.  Lets a trainer's epoch log line (loss, source
    micro-F1) ride on the loss kernel instead of an argmax / compare / sum / cast / stack chain."""
    global _ce_stats
    hit, _ce_stats = _ce_stats, None              # consumed by the first look-up: a later loss computed some other
    if hit is None:                                                                         # way never sees it
        return None
    lg, lb = hit[0](), hit[1]()
    if (lg is not None and lb is not None and hit[2] == logits.size(0) and lg.shape == logits.shape
            and lg.data_ptr() == logits.data_ptr() and lb.data_ptr() == labels.data_ptr()
            and lg._version == hit[4] and lb._version == hit[5]):
        return hit[3]
    return None


def softmax_nll(logits, labels):
    """``F.nll_loss(F.log_softmax(logits, dim=1), labels)`` (mean over rows) in one pass each way."""
    return _SoftmaxNLL.apply(logits, labels)


def source_ce(logits, labels):
    """The trainers' source loss line (pygda/models/a2gnn.py:182): fused kernel for device logits with
    up to 64 classes, the torch composition otherwise."""
    if logits.is_cuda and logits.dim() == 2 and logits.size(1) <= 64 and logits.dtype == torch.float32:
        return softmax_nll(logits, labels)
    import torch.nn.functional as F
    return F.nll_loss(F.log_softmax(logits, dim=1), labels)


class _SoftmaxEntropy(torch.autograd.Function):
    """``mean_i sum_c -q log q``, ``q = clamp(softmax(logits_i), lo, 1)`` (gda_softmax_entropy_{fwd,bwd}_f32)."""

    @staticmethod
    def forward(ctx, logits, lo):
        x = _f32c(logits, "logits")
        n, c = x.shape
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws = _lib.workspace(L.gda_softmax_nll_workspace_bytes(), x.device, "entropy")
        _lib.check(L.gda_softmax_entropy_fwd_f32(_lib.ptr(x), c, n, c, float(lo), _lib.ptr(loss), _lib.ptr(ws), ws.numel(),
                                                 _lib.stream()), "gda_softmax_entropy_fwd_f32")
        ctx.save_for_backward(x)
        ctx.lo = float(lo)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        (x,) = ctx.saved_tensors
        n, c = x.shape
        gx = torch.empty_like(x)
        gl = gl.reshape(1).to(torch.float32).contiguous()
        _lib.check(_lib.lib().gda_softmax_entropy_bwd_f32(_lib.ptr(x), c, n, c, ctx.lo, _lib.ptr(gl), _lib.ptr(gx), c,
                                                         _lib.stream()), "gda_softmax_entropy_bwd_f32")
        return gx, None


def softmax_entropy(logits, clamp_min=1e-9):
    """UDAGCN's target entropy term (pygda/models/udagcn.py:193-197): ``p = clamp(softmax(logits, -1), min, 1.0)``,
    ``mean(sum(-p log p, -1))`` -- one row kernel each way for device logits with up to 64 classes, the reference's
    composition otherwise."""
    if (logits.is_cuda and logits.dim() == 2 and 0 < logits.size(1) <= 64 and logits.size(0) > 0
            and logits.dtype == torch.float32):
        return _SoftmaxEntropy.apply(logits, clamp_min)
    import torch.nn.functional as F
    p = torch.clamp(F.softmax(logits, dim=-1), min=clamp_min, max=1.0)
    return torch.mean(torch.sum(-p * torch.log(p), dim=-1))


# --------------------------------------------------------- tall-skinny GEMMs (MFMA) --
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2


def gemm_raw(mode, M, N, K, a, lda, b, ldb, c, ldc, name="dense_projection"):
    """gda_gemm_f32 on explicit leading dimensions (column-major activations are row-major matrices of the
    transposed shape with a padded leading dimension)."""
    L = _lib.lib()
    need = L.gda_gemm_workspace_bytes(mode, M, N, K)
    ws = _lib.workspace(need, c.device, "gemm") if need else None
    with profiler.region(name, 1, 4 * (M * K + N * K + M * N), 2 * M * N * K):
        _lib.check(L.gda_gemm_f32(mode, M, N, K, _lib.ptr(a), lda, _lib.ptr(b), ldb, _lib.ptr(c), ldc,
                                  _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream()),
                   "gda_gemm_f32")
    return c


TALL_ROWS = int(_os.environ.get("PYGDA_AMD_TALL_GEMM_ROWS", "32768"))    # node counts from here on: the tall kernels


# the weight gradient gy^T x with one workgroup per row slab (k_tall_wgrad) from 4 k rows on: at cfg-A's stacked source rows
# (18,720) the 64 x 64-tile kernel's 64 slabs took 27 us on the tail of the step, the slab kernel 2 - 3 % off the step; the
# target's 5,484 rows give another 2 % (same box, ms/step: 32768 -> 0.4355 / 0.4338, 8192 -> 0.4271 / 0.4348, 4096 -> 0.4252 / 0.4128)
TALL_WGRAD_ROWS = int(_os.environ.get("PYGDA_AMD_TALL_WGRAD_ROWS", "4096"))


def _tall_shape(mode, M, N, K, a, b):
    """The envelope of gda_gemm_tall_f32 (csrc/gda_gemm.hip): sampled sub-graphs (10^5 rows and more) against a weight
    whose extents are 128 or 256."""
    if mode == GEMM_TN:          # gW = gy^T x: `a` = gy [rows, 128], reduction over the rows; 16-byte row loads of both
        return (M == 128 and N in (128, 256) and K >= TALL_WGRAD_ROWS and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)
    return M >= TALL_ROWS and N in (128, 256) and K in (128, 256) and a.data_ptr() % 16 == 0


SKINNY_ROWS = int(_os.environ.get("PYGDA_AMD_SKINNY_GEMM_ROWS", "2048"))   # the classifier projection's vector kernels from here on


def _skinny_shape(mode, M, N, K, a, b, c_ld):
    """The envelope of gda_gemm_skinny_f32: the classifier projection (at most 8 classes).  Round 5: from citation size
    on, not only at sampled-batch row counts -- a 64 x 64 matrix-core tile is 92 % padding for 5 classes, and on the
    64 x 64-tile kernel the classifier's weight gradient (9,360 rows -> a 5 x 128 result) was a 22 us kernel on the tail of
    the cfg-A step."""
    wide = (32, 64, 128, 256)
    al = lambda t: t.data_ptr() % 16 == 0
    if mode == GEMM_NT:
        return M >= SKINNY_ROWS and N <= 8 and K in wide and al(a)
    if mode == GEMM_NN:
        return M >= SKINNY_ROWS and K <= 8 and N in wide and al(b)
    return K >= SKINNY_ROWS and M <= 8 and N in wide and al(b)


def gemm(mode, a, b, bias=None, colsum=None):
    """fp32 product on the matrix cores (include/gda_hip.h: gda_gemm_ex_f32 / gda_gemm_tall_f32), no autograd.
    NT: ``a [M,K] @ b [N,K]^T`` (+ ``bias [N]`` in the epilogue); NN: ``a [M,K] @ b [K,N]``;
    TN: ``a [K,M]^T @ b [K,N]`` (``colsum [M]`` receives ``a.sum(0)``, the bias gradient beside the weight's)."""
    a, b = _f32c(a, "a"), _f32c(b, "b")
    if mode == GEMM_NT:
        (M, K), (N, K2) = a.shape, b.shape
    elif mode == GEMM_NN:
        (M, K), (K2, N) = a.shape, b.shape
    else:
        (K, M), (K2, N) = a.shape, b.shape
    if K != K2:
        raise ValueError(f"inner dimensions differ: {tuple(a.shape)} x {tuple(b.shape)} (mode {mode})")
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    L = _lib.lib()
    tall = _tall_shape(mode, M, N, K, a, b)
    skinny = not tall and _skinny_shape(mode, M, N, K, a, b, N)
    need = (L.gda_gemm_tall_workspace_bytes if tall else L.gda_gemm_skinny_workspace_bytes if skinny
            else L.gda_gemm_workspace_bytes)(mode, M, N, K)
    ws = _lib.workspace(need, a.device, "gemm") if need else None
    name = ("dense_projection", "dense_projection_dgrad", "dense_projection_wgrad")[mode]
    with profiler.region(f"{name}[{K}x{N}]" if mode != GEMM_TN else f"{name}[{M}x{N}]", 1,
                         4 * (a.numel() + b.numel() + c.numel()), 2 * M * N * K):
        fn, what = ((L.gda_gemm_tall_f32, "gda_gemm_tall_f32") if tall else
                    (L.gda_gemm_skinny_f32, "gda_gemm_skinny_f32") if skinny else (L.gda_gemm_ex_f32, "gda_gemm_ex_f32"))
        _lib.check(fn(mode, M, N, K, _lib.ptr(a), a.size(1), _lib.ptr(b), b.size(1), _lib.ptr(c), N,
                      _lib.ptr(bias), _lib.ptr(colsum), _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream()),
                   what)
    return c


# --------------------------------------------------------------------------- SpMM --
aggregated_edges = 0     # running count of nnz(A_hat) over every aggregation launched (bench bookkeeping;
                         # only maintained while the profiler is on: it costs a cached-nnz lookup)
aggregation_log = None   # or a list: (graph or _Nnz, K) per aggregation call, nnz resolved later (no sync in the loop)


class _Nnz:
    """What the log keeps of a graph whose entry count is already known on the host (every batch of the device
    sampler): the number, not the graph -- a logged graph object pins its batch's CSR buffers for as long as the log
    lives, ~13 MB per sampled step that the caching allocator then has to hipMalloc afresh (1.7 device allocations per
    step inside bench.py's timed cfg-S region, one of them now and then 30 ms long)."""
    __slots__ = ("nnz", "tag")

    def __init__(self, nnz, tag=None):
        self.nnz, self.tag = nnz, tag


def _logged(graph):
    return (_Nnz(graph._nnz, getattr(graph, "tag", None))
            if getattr(graph, "_nnz", None) is not None and getattr(graph, "transient", False) else graph)

kstep_paths = None       # or a dict: which kernel ran the K >= 3 aggregation calls ("lds-one-launch" / "launch-chain"),
                         # counted per call while the profiler is on (bench.py: config.kstep_aggregation_path)


def _note_path(name, K):
    if kstep_paths is not None and profiler.enabled and K >= 3:
        kstep_paths[name] = kstep_paths.get(name, 0) + 1


INTERIOR_KSTEP = _os.environ.get("PYGDA_AMD_INTERIOR_KSTEP", "1") == "1"


INTERIOR_HOIST = _os.environ.get("PYGDA_AMD_INTERIOR_HOIST", "1") == "1"


def _launch_kstep_interior(graph, x, K, bias, transposed, y):
    """K steps on a sampled batch whose rows ``[n_interior, n)`` hold their unit self loop only: the interior rows are
    recomputed per step, the leaves are finished in one pass (csrc/gda_spmm.hip, "sampled sub-graphs")."""
    rp, ci, va = (graph.t_rowptr, graph.t_colidx, graph.t_val) if transposed else (graph.rowptr, graph.colidx, graph.val)
    n, d = x.shape
    n_int = graph.n_interior
    L = _lib.lib()
    plan = _interior_lds_plan(graph, x, y, K, transposed)
    if aggregation_log is not None:
        # (graph, K) = the K full aggregations the call stands for (SURVEY 8d's reference-equivalent count); third
        # element = the entries whose multiply-add this call really EXECUTES.  One-launch path: every entry of the batch
        # once (the leaf columns' contribution / the leaves' final pass, the leaf rows' unit self loops as one copy) + the
        # T off-diagonal and n_int diagonal entries of the interior block in each of the other K - 1 steps.  Launch
        # chain: its K steps walk every entry of the interior rows (skipped ones stay in the chain as w * 0).
        nnz_h, T = getattr(graph, "_nnz", None), getattr(graph, "iplan_T", None)
        executed = None
        if nnz_h is not None:
            executed = (nnz_h + (int(K) - 1) * (T + n_int) if (plan is not None and T is not None)
                        else int(K) * (nnz_h - (n - n_int)) + (n - n_int))
        aggregation_log.append((_logged(graph), int(K), executed, "interior-lds" if plan is not None else "interior-rows"))
    _note_path("interior-lds" if plan is not None else "interior-rows", int(K))
    if profiler.enabled:
        # `bytes`: what the call itself has to move -- forward K interior steps + one copy of the leaf rows; transposed K
        # interior steps + one pass over every entry for the leaves.  `alg_equiv_bytes`: SURVEY 8(d)'s algorithmic
        # bytes of the K full aggregations the call stands for.
        global aggregated_edges
        aggregated_edges += int(K) * graph.nnz
        nnz, n_leaf, row = graph.nnz, n - n_int, 4 * d
        # compulsory bytes: a step reads every distinct row its entries name at most once (<= n rows; the transposed
        # interior steps name interior rows only), the index arrays, and writes its own rows
        hoist = INTERIOR_HOIST and not transposed and K >= 2 and n_int > 0
        alg = K * (nnz * 8 + (n + 1) * 4 + 2 * n * d * 4)
        if plan is not None:
            # the one-launch path: the step loop never leaves LDS; HBM / L2 see the index arrays once, the leaf rows
            # once (forward: gathered for c; transposed: written), and the interior block crossing the column-major
            # staging (in: x_I and c or g_I; out: y_I, and the summed step inputs when transposed)
            real = nnz * 8 + (min(nnz, n) if not transposed else n_int) * row + 6 * n_int * row + 2 * n_leaf * row
            ctx = profiler.region(f"interior_lds_f32[d={d}]", 4 if transposed else 3, real, K * 2 * nnz * d,
                                  alg_equiv_bytes=alg, step_loop_launches=1)
        else:
            if transposed:
                real = K * (3 * n_int * row) + nnz * 8 + n_int * row + 2 * n_leaf * row
            elif hoist:   # one pass over the leaf columns, then K steps that gather interior rows and add the constant
                real = (nnz * 8 + min(nnz, n) * row + n_int * row) + K * (nnz * 8 + 3 * n_int * row + (n_int + 1) * 4) \
                    + 2 * n_leaf * row
            else:
                real = K * (nnz * 8 + min(nnz, n) * row + n_int * row + (n_int + 1) * 4) + 2 * n_leaf * row
            # launches: forward = K (+ 1 with the hoisted leaf term) of k_spmm_range + ONE k_rows_copy_bias; transposed =
            # K + 1 of k_spmm_range (`copy_launches` lets the bench price the call with rocprofv3's per-kernel averages)
            ctx = profiler.region(f"spmm_interior_f32[d={d}]", K + (2 if hoist else 1), real, K * 2 * nnz * d,
                                  alg_equiv_bytes=alg, copy_launches=0 if transposed else 1)
    else:
        ctx = profiler.region("", 0)
    if plan is not None:
        # ONE launch for the K steps (csrc/gda_interior.inc): 3 launches per call forward, 4 transposed, whatever K is
        # sized for the LARGEST interior block the kernel takes, not for this batch's: a scratch that grows with every
        # record-size batch is a hipMalloc (tens of milliseconds) in the middle of a training step
        nbytes = L.gda_interior_kstep_lds_workspace_bytes(_interior_lds_limits()[0], d)
        ws = _lib.workspace(nbytes, x.device, "interior_lds")
        with ctx:
            _lib.check(L.gda_interior_kstep_lds_f32(_lib.ptr(rp), _lib.ptr(ci), _lib.ptr(va), n, n_int, d, int(K),
                                                    int(bool(transposed)), _lib.ptr(plan), _lib.ptr(x), _lib.ptr(y),
                                                    _lib.ptr(bias), _lib.ptr(ws), nbytes, _lib.stream()),
                       "gda_interior_kstep_lds_f32")
        return
    tmp = torch.empty(n_int, d, dtype=torch.float32, device=x.device) if K > 1 and n_int else None
    # transposed: the running sum of the step inputs; forward (K >= 2): the leaf columns' contribution, formed once
    want_sacc = transposed or (INTERIOR_HOIST and K >= 2)
    sacc = torch.empty(n_int, d, dtype=torch.float32, device=x.device) if want_sacc and n_int else None
    with ctx:
        _lib.check(L.gda_spmm_csr_interior_kstep_f32(_lib.ptr(rp), _lib.ptr(ci), _lib.ptr(va), n, n_int, d, int(K),
                                                     int(bool(transposed)), _lib.ptr(x), _lib.ptr(y), _lib.ptr(tmp),
                                                     _lib.ptr(sacc), _lib.ptr(bias), _lib.stream()),
                   "gda_spmm_csr_interior_kstep_f32")


INTERIOR_LDS_MIN_K = int(_os.environ.get("PYGDA_AMD_INTERIOR_LDS_MIN_K", "3"))


def _interior_lds_plan(graph, x, y, K, transposed):
    """The device plan of this direction when the K interior steps of a sampled batch run as one launch
    (csrc/gda_interior.inc), else None: the device sampler built a valid plan for the direction (sampler.INTERIOR_LDS),
    the width is one the kernel takes (a multiple of 4, one workgroup per column up to 128), K is worth it."""
    plans = getattr(graph, "iplan", None)
    if plans is None or K < INTERIOR_LDS_MIN_K or not INTERIOR_HOIST:      # hoist off = the chain's bit-exact row sums asked for
        return None
    plan = plans[1 if transposed else 0]
    n_int, d = graph.n_interior, x.size(1)
    if plan is None or not n_int or d % 4 or d > _interior_lds_limits()[1] or n_int > _interior_lds_limits()[0]:
        return None
    if x.data_ptr() % 16 or y.data_ptr() % 16:
        return None
    return plan


_il_limits = None


def _interior_lds_limits():
    global _il_limits
    if _il_limits is None:
        L = _lib.lib()
        _il_limits = (int(L.gda_interior_max_rows()), int(L.gda_interior_max_width()))
    return _il_limits


def _takes_interior_path(graph, x, bias, transposed, counts_as=None):
    """The ONE predicate for "these K steps run on the interior-rows kernel" (it needs no ping-pong buffer):
    used by the launcher and by every caller that decides whether to allocate one."""
    return bool(INTERIOR_KSTEP and counts_as is None and graph.n_interior is not None
                and 2 * graph.n_interior <= x.size(0) and x.is_contiguous() and (bias is None or not transposed))


def _launch_kstep(graph, x, K, bias, transposed, y, tmp, counts_as=None):
    """``counts_as = (graph, steps)``: what the launches stand for in the edges-aggregated bookkeeping
    (a launch of the cached A*A counts as two aggregations over the edges of A, not over its own)."""
    if _takes_interior_path(graph, x, bias, transposed, counts_as):
        return _launch_kstep_interior(graph, x, K, bias, transposed, y)
    if tmp is None and K > 1:          # a caller that expected the interior path (which needs no ping-pong buffer)
        tmp = torch.empty_like(x)
    rp, ci, va = (graph.t_rowptr, graph.t_colidx, graph.t_val) if transposed else \
                 (graph.rowptr, graph.colidx, graph.val)
    n, d = x.shape
    L = _lib.lib()
    book_graph, book_steps = counts_as if counts_as is not None else (graph, int(K))
    _note_path("launch-chain", int(K))
    if aggregation_log is not None:
        aggregation_log.append((_logged(book_graph), book_steps))
    if profiler.enabled:      # algorithmic bytes per launch: nnz*(4+4) + (N+1)*4 + 2*N*d*4
        global aggregated_edges
        aggregated_edges += book_steps * book_graph.nnz
        nbytes = K * (graph.nnz * 8 + (n + 1) * 4 + 2 * n * d * 4)
        ctx = profiler.region(f"spmm_csr_f32[d={d}]", K, nbytes, K * 2 * graph.nnz * d)
    else:
        ctx = profiler.region("", 0)
    sp = graph.split(transposed).struct(d)         # None unless the graph has hub rows (> SPLIT_THRESHOLD entries)
    with ctx:
        _lib.check(L.gda_spmm_csr_split_f32(_lib.ptr(rp), _lib.ptr(ci), _lib.ptr(va), n, d, int(K),
                                            _lib.ptr(x), d, _lib.ptr(y), d, _lib.ptr(tmp), _lib.ptr(bias),
                                            ctypes.byref(sp) if sp is not None else None,
                                            _lib.stream()), "gda_spmm_csr_split_f32")


def _launch_kstep_lds(graph, plan, slots, x, K, bias, transposed, y, x_colmajor=False, y_colmajor=False,
                      colsum=None):
    """All K steps in one launch on the LDS-resident kernel (csrc/gda_kstep.hip): same sums, bit for bit.
    Column-major operands are ``[d, round_up(n, 4)]`` tensors and skip the transposition on their side."""
    n = graph.num_nodes
    d = x.size(0) if x_colmajor else x.size(1)
    L = _lib.lib()
    if aggregation_log is not None:
        aggregation_log.append((graph, int(K)))
    if profiler.enabled:
        global aggregated_edges
        aggregated_edges += int(K) * graph.nnz
        _note_path("lds-one-launch", int(K))
        # `bytes`: SURVEY 8(d)'s algorithmic bytes of the K aggregations the launch stands for; `hbm_bytes`: what the
        # kernel itself moves (the plan per workgroup + one read and one write of the activations); `lds_bytes`:
        # the words its step loops gather out of LDS (slot-program entries incl. padding x columns x K)
        ctx = profiler.region(f"kstep_lds_f32[d={d},K={int(K)}]", 1,
                              K * (graph.nnz * 8 + (n + 1) * 4 + 2 * n * d * 4), K * 2 * graph.nnz * d,
                              hbm_bytes=plan.numel() + 2 * n * d * 4,
                              lds_bytes=4 * 1024 * (slots & 0xff) * 4 * d * int(K))
    else:
        ctx = profiler.region("", 0)
    n_pad = (n + 3) // 4 * 4
    ws = _lib.workspace(2 * d * n_pad * 4, x.device, "kstep")
    if profiler.enabled:
        # the same three launches as separate C calls, so that the K-step kernel itself is bracketed by the
        # HIP events (roofline: algorithmic bytes of K aggregations / its own duration)
        wsf = ws.view(torch.float32)
        xT, yT = (x if x_colmajor else wsf[:d * n_pad]), (y if y_colmajor else wsf[d * n_pad:2 * d * n_pad])
        if not x_colmajor:
            with profiler.region(f"transpose[{n}x{d}]", 1, 8 * n * d, 0):
                _lib.check(L.gda_transpose_f32(_lib.ptr(x), d, _lib.ptr(xT), n_pad, n, d, _lib.stream()),
                           "gda_transpose_f32")
        with ctx:
            _lib.check(L.gda_kstep_lds_colmajor_f32(_lib.ptr(plan), slots, n, d, int(K), _lib.ptr(xT), n_pad,
                                                    _lib.ptr(yT), n_pad, _lib.ptr(bias), _lib.ptr(colsum),
                                                    _lib.stream()), "gda_kstep_lds_colmajor_f32")
        if not y_colmajor:
            with profiler.region(f"transpose[{n}x{d}]", 1, 8 * n * d, 0):
                _lib.check(L.gda_transpose_f32(_lib.ptr(yT), n_pad, _lib.ptr(y), d, d, n, _lib.stream()),
                           "gda_transpose_f32")
        return
    _lib.check(L.gda_kstep_lds_f32(_lib.ptr(plan), slots, n, d, int(K),
                                   _lib.ptr(x), n_pad if x_colmajor else d, int(x_colmajor),
                                   _lib.ptr(y), n_pad if y_colmajor else d, int(y_colmajor),
                                   _lib.ptr(bias), _lib.ptr(colsum), _lib.ptr(ws), _lib.stream()),
               "gda_kstep_lds_f32")


def spmm_kstep(graph: CSRGraph, x, K=1, bias=None, transposed=False):
    """``A_hat^K @ x (+ bias)`` without autograd (ping-pong buffers), K launches.

    Opt-in (``PYGDA_AMD_SQUARE=1``): for the STATIC graph of a full-batch loader, K // 2 launches of
    the cached ``A_hat * A_hat`` plus K % 2 of ``A_hat`` -- at citation-graph sizes a dependent launch
    costs its latency (5-7 us), not its edges.  Same product up to fp32 summation order; measured
    +4 % epochs/s at cfg-A, at the price of the bit-exact edge-order sums, so it is off by default."""
    from .graph import KSTEP_LDS, KSTEP_LDS_MIN_K, SQUARE
    x = _f32c(x, "x")
    if x.dim() != 2 or x.size(0) != graph.num_nodes:
        raise ValueError(f"x must be [num_nodes={graph.num_nodes}, d], got {tuple(x.shape)}")
    y = torch.empty_like(x)
    b = None if bias is None else _f32c(bias, "bias")
    sq = graph.squared() if (SQUARE and K >= 2 and graph.static) else None
    if sq is None and KSTEP_LDS and K >= KSTEP_LDS_MIN_K and graph.static:
        hit = graph.kstep_plan(transposed)       # static full-batch graphs at citation size: one launch
        if hit is not None:
            _launch_kstep_lds(graph, hit[0], hit[1], x, K, b, transposed, y)
            return y
    if sq is None:
        interior = _takes_interior_path(graph, x, b, transposed)
        _launch_kstep(graph, x, K, b, transposed, y, torch.empty_like(x) if K > 1 and not interior else None)
        return y
    pairs, single = K // 2, K % 2
    if single:
        mid = torch.empty_like(x)
        _launch_kstep(graph, x, 1, None, transposed, mid, None)
        x = mid
    _launch_kstep(sq, x, pairs, b, transposed, y, torch.empty_like(x) if pairs > 1 else None,
                  counts_as=(graph, 2 * pairs))
    return y


class ColMajor:
    """A ``[n, d]`` activation held column-major as ``t [d, round_up(n, 4)]`` -- the layout the LDS-resident
    K-step kernel computes in.  Produced by :func:`propagate` (``colmajor_out=True``) for its one consumer,
    :func:`relu_dropout`, whose kernel transposes on the fly; ``dense()`` gives the ordinary tensor."""
    __slots__ = ("t", "n")

    def __init__(self, t, n):
        self.t, self.n = t, n

    def detach(self):
        return ColMajor(self.t.detach(), self.n)

    @property
    def shape(self):
        return torch.Size((self.n, self.t.size(0)))

    def size(self, k=None):
        return self.shape if k is None else self.shape[k]

    def dense(self):
        return self.t[:, :self.n].t().contiguous()


def lds_kstep_plan(graph, K, transposed=False):
    """``(plan, slots)`` when ``K`` aggregation steps over ``graph`` run on the one-launch LDS kernel."""
    from .graph import KSTEP_LDS, KSTEP_LDS_MIN_K, SQUARE
    if not KSTEP_LDS or SQUARE or K < KSTEP_LDS_MIN_K or not graph.static:
        return None
    return graph.kstep_plan(transposed)


class _TallLinearT(torch.autograd.Function):
    """``h = x W^T`` handed over COLUMN-MAJOR (``hT [out, round_up(n, 4)]``) to the K-step kernel: the same
    matrix-core kernels as ``nn.linear._TallLinear`` with the operands swapped -- ``hT = W x^T`` is the NT
    product of the small matrix with the tall one --, and in the backward pass ``gx = gT^T W`` (TN) and
    ``gW = gT x`` (NN with the deterministic row-slab split) straight from the column-major gradient."""

    @staticmethod
    def forward(ctx, x, weight):
        x, weight = _f32c(x, "x"), _f32c(weight, "weight")
        n, k = x.shape
        out = weight.size(0)
        n_pad = (n + 3) // 4 * 4
        hT = torch.empty(out, n_pad, dtype=torch.float32, device=x.device)
        gemm_raw(GEMM_NT, out, n, k, weight, k, x, k, hT, n_pad, f"dense_projection[{k}x{out}]")
        ctx.save_for_backward(x, weight)
        return hT

    @staticmethod
    def backward(ctx, gT):
        x, weight = ctx.saved_tensors
        n, k = x.shape
        out = weight.size(0)
        gT = gT.contiguous()
        n_pad = gT.size(1)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(n, k, dtype=torch.float32, device=x.device)
            gemm_raw(GEMM_TN, n, k, out, gT, n_pad, weight, k, gx, k, f"dense_projection_dgrad[{out}x{k}]")
        if ctx.needs_input_grad[1]:
            gw = torch.empty(out, k, dtype=torch.float32, device=x.device)
            gemm_raw(GEMM_NN, out, k, n, gT, n_pad, x, k, gw, k, f"dense_projection_wgrad[{out}x{k}]")
        return gx, gw


def tall_linear_colmajor(x, weight):
    return ColMajor(_TallLinearT.apply(x, weight), x.size(0))


_colsum_hint = None      # (data_ptr, column sums, the tensor itself) left by a kernel that produced both


def colsum(x):
    """``x.sum(0)`` of a row-major ``[n, d]`` fp32 device matrix (bias gradients): deterministic two-stage sum --
    or the by-product of the kernel that just wrote ``x`` (the stacked activation's backward), when there is one."""
    global _colsum_hint
    hint, _colsum_hint = _colsum_hint, None
    if hint is not None and hint[0] == x.data_ptr() and hint[2].shape == x.shape and x.is_contiguous():
        return hint[1]
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 0 < x.size(1) <= 1024 and x.stride(1) == 1):
        return x.sum(0)
    n, d = x.shape
    out = torch.empty(d, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    nbytes = L.gda_colsum_workspace_bytes(n, d)
    ws = _lib.workspace(nbytes, x.device, "colsum")
    _lib.check(L.gda_colsum_f32(_lib.ptr(x), x.stride(0), n, d, _lib.ptr(out), _lib.ptr(ws), nbytes, _lib.stream()),
               "gda_colsum_f32")
    return out


class _PropagateT(torch.autograd.Function):
    """K-step aggregation with a COLUMN-MAJOR result (forward) and gradient (backward): the transposes on
    the activation side of the LDS kernel disappear into the fused activation kernels
    (``gda_relu_dropout_*_cm_f32``), and the bias gradient is the column sum the backward kernel takes of its
    input while loading it -- no reduction launch."""

    @staticmethod
    def forward(ctx, x, bias, graph, K, x_colmajor=False):
        """``x``: ``[n, d]`` row-major, or (``x_colmajor``) ``[d, round_up(n, 4)]`` column-major -- then the
        gradient goes back column-major as well and no transposition is left on either side."""
        x = _f32c(x, "x")
        n = graph.num_nodes
        d = x.size(0) if x_colmajor else x.size(1)
        plan, slots = graph.kstep_plan(False)
        n_pad = (n + 3) // 4 * 4
        yT = torch.empty(d, n_pad, dtype=torch.float32, device=x.device)
        b = None if bias is None else _f32c(bias, "bias")
        _launch_kstep_lds(graph, plan, slots, x, K, b, False, yT, x_colmajor=x_colmajor, y_colmajor=True)
        ctx.graph, ctx.K, ctx.has_bias, ctx.n, ctx.x_colmajor = graph, K, bias is not None, n, x_colmajor
        return yT

    @staticmethod
    def backward(ctx, gT):
        graph, n = ctx.graph, ctx.n
        gT = gT.contiguous()
        d, n_pad = gT.shape
        plan, slots = graph.kstep_plan(True)
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        gb = torch.empty(d, dtype=torch.float32, device=gT.device) if want_b else None
        if ctx.x_colmajor:
            gx = torch.empty(d, n_pad, dtype=torch.float32, device=gT.device)
            if n_pad != n:
                gx[:, n:].zero_()
        else:
            gx = torch.empty(n, d, dtype=torch.float32, device=gT.device)
        _launch_kstep_lds(graph, plan, slots, gT, ctx.K, None, True, gx, x_colmajor=True,
                          y_colmajor=ctx.x_colmajor, colsum=gb)
        return (gx if ctx.needs_input_grad[0] else None), gb, None, None, None


class _Propagate(torch.autograd.Function):
    """K-step neighbour aggregation.  Linear in x, so nothing but the graph is saved:
    backward is the same K-step kernel on the by-source CSR (A_hat^T)."""

    @staticmethod
    def forward(ctx, x, bias, graph, K):
        ctx.graph, ctx.K, ctx.has_bias = graph, K, bias is not None
        return spmm_kstep(graph, x, K, bias)

    @staticmethod
    def backward(ctx, gy):
        gx = spmm_kstep(ctx.graph, gy.contiguous(), ctx.K, None, transposed=True) \
            if ctx.needs_input_grad[0] else None
        gb = colsum(gy) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gx, gb, None, None


def propagate(x, graph: CSRGraph, K=1, bias=None, colmajor_out=False):
    """``out = A_hat^K x + bias`` with autograd (prop_gcn_conv.py:208-213).  ``colmajor_out``: hand the result
    to the fused activation as a :class:`ColMajor` when the LDS-resident kernel runs this call (static graph
    at citation size, K >= 3, width a multiple of 4) -- otherwise the ordinary tensor comes back."""
    if K < 1:
        raise ValueError("K must be >= 1")
    if isinstance(x, ColMajor):              # from tall_linear_colmajor(): eligibility was checked there
        return ColMajor(_PropagateT.apply(x.t, bias, graph, int(K), True), x.n)
    if colmajor_out and lds_colmajor_ok(x, graph, K):
        return ColMajor(_PropagateT.apply(x, bias, graph, int(K)), x.size(0))
    return _Propagate.apply(x, bias, graph, int(K))


FUSED_INTERIOR_ACT = _os.environ.get("PYGDA_AMD_FUSED_INTERIOR_ACT", "1") == "1"


def propagate_act_ok(x, graph, K, bias):
    """``relu_dropout(propagate(x, graph, K, bias))`` runs as the one-launch interior K-step with the activation in its
    epilogue (gda_interior_kstep_lds_act_f32): a sampled batch whose forward plan the device sampler built."""
    return (FUSED_INTERIOR_ACT and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and x.is_contiguous() and _takes_interior_path(graph, x, bias, False)
            and _interior_lds_plan(graph, x, x, int(K), False) is not None)


class _PropagateAct(torch.autograd.Function):
    """``act(A_hat^K x + bias)`` with ``act = dropout(relu(.))`` applied by the aggregation's own epilogue; with ``pair`` a
    second output carries an independent dropout draw of the same pre-activation (not differentiated: the trainer's
    loss-unused pass).  Backward: the activation's mask from the saved OUTPUT (y > 0 <=> x > 0 and kept), then the
    K-step kernel on the by-source CSR -- nothing but the output and the graph is saved."""

    @staticmethod
    def forward(ctx, x, bias, graph, K, p, pair):
        x = _f32c(x, "x")
        n, d = x.shape
        n_int = graph.n_interior
        y0 = torch.empty_like(x)
        y1 = torch.empty_like(x) if pair else None
        b = None if bias is None else _f32c(bias, "bias")
        plan = _interior_lds_plan(graph, x, y0, K, False)
        st = dropout_state
        if st.seed is None:
            st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        site0 = st.next_site()
        site1 = st.next_site() if pair else 0
        L = _lib.lib()
        if aggregation_log is not None:
            nnz_h, T = getattr(graph, "_nnz", None), getattr(graph, "iplan_T", None)
            executed = None
            if nnz_h is not None:
                executed = (nnz_h + (int(K) - 1) * (T + n_int) if T is not None
                            else int(K) * (nnz_h - (n - n_int)) + (n - n_int))
            aggregation_log.append((_logged(graph), int(K), executed, "interior-lds"))
        _note_path("interior-lds", int(K))
        if profiler.enabled:
            global aggregated_edges
            aggregated_edges += int(K) * graph.nnz
            nnz, n_leaf, row = graph.nnz, n - n_int, 4 * d
            alg = K * (nnz * 8 + (n + 1) * 4 + 2 * n * d * 4)
            real = nnz * 8 + min(nnz, n) * row + 6 * n_int * row + (2 + bool(pair)) * n_leaf * row
            region = profiler.region(f"interior_lds_f32[d={d}]", 3, real, K * 2 * nnz * d, alg_equiv_bytes=alg,
                                     step_loop_launches=1)
        else:
            region = profiler.region("", 0)
        nbytes = L.gda_interior_kstep_lds_workspace_bytes(_interior_lds_limits()[0], d)
        ws = _lib.workspace(nbytes, x.device, "interior_lds")
        with region:
            _lib.check(L.gda_interior_kstep_lds_act_f32(
                _lib.ptr(graph.rowptr), _lib.ptr(graph.colidx), _lib.ptr(graph.val), n, n_int, d, int(K), _lib.ptr(plan),
                _lib.ptr(x), _lib.ptr(y0), _lib.ptr(y1), _lib.ptr(b), float(p), ctypes.c_uint64(st.seed),
                _lib.ptr(st.counter(x.device)), ctypes.c_uint32(site0), ctypes.c_uint32(site1), _lib.ptr(ws), nbytes,
                _lib.stream()), "gda_interior_kstep_lds_act_f32")
        ctx.graph, ctx.K, ctx.has_bias, ctx.p = graph, K, bias is not None, float(p)
        ctx.save_for_backward(y0)
        ctx.sink = _offer_sink(y0, p, wanted=ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        if pair:
            ctx.mark_non_differentiable(y1)
            return y0, y1
        return y0

    @staticmethod
    def backward(ctx, g0, g1=None):
        (y0,) = ctx.saved_tensors
        if ctx.sink is not None and ctx.sink.mine(g0):       # masked by its producer already (GradSink)
            gpre = g0
        else:
            g0 = g0.contiguous()
            gpre = torch.empty_like(g0)
            _lib.check(_lib.lib().gda_relu_dropout_bwd_f32(_lib.ptr(g0), _lib.ptr(y0), _lib.ptr(gpre), g0.numel(), ctx.p,
                                                           _lib.stream()), "gda_relu_dropout_bwd_f32")
        gx = spmm_kstep(ctx.graph, gpre, ctx.K, None, transposed=True) if ctx.needs_input_grad[0] else None
        gb = colsum(gpre) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gx, gb, None, None, None, None


def propagate_act(x, graph, K, bias, p, training, pair=False):
    """``relu_dropout(propagate(x, graph, K, bias), p, training)`` -- twice, with independent draws, when ``pair`` -- as ONE
    call when :func:`propagate_act_ok`; the caller checks that first."""
    return _PropagateAct.apply(x, bias, graph, int(K), float(p) if training else 0.0, bool(pair))


def lds_colmajor_ok(x, graph, K):
    """The K-step call on ``x [n, d]`` runs on the LDS kernel and may hand over a column-major result."""
    return (x.dim() == 2 and x.size(1) % 4 == 0 and x.is_cuda and x.dtype == torch.float32
            and lds_kstep_plan(graph, int(K), False) is not None and lds_kstep_plan(graph, int(K), True) is not None)


# ---------------------------------------------------------------------------- MMD --
MMD_INDEX_IN_KERNEL = _os.environ.get("PYGDA_AMD_MMD_INDEX", "0") == "1"     # sampled rows read through their index inside the kernels (no gather pass)
MMD_SCATTER_FUSED = _os.environ.get("PYGDA_AMD_MMD_SCATTER", "1") == "1"
# loss + unscaled row gradients in one pass over the pairs on the 16-bit matrix cores with split operands
# (csrc/gda_mmd_fused.inc); 0 = the two-pass fp32-MFMA kernels with the [times, m, m] weight matrix in between
MMD_ONE_PASS = _os.environ.get("PYGDA_AMD_MMD_ONE_PASS", "1") == "1"


# the chunked one-pass kernel (csrc/gda_mmd_chunked.inc: any width up to 1024): "auto" = the widths the register-resident
# kernel does not cover (d > 128 or not a multiple of 32: GRADE's 645), "always" = every width (experiments), "0" = never
MMD_CHUNKED = _os.environ.get("PYGDA_AMD_MMD_CHUNKED", "auto")


def mmd_chunked_plan(times, n, d, kernel_mul=2.0, kernel_num=5):
    """``(nseg, padded width)`` when the chunked one-pass MMD takes these shapes, else None."""
    if not MMD_ONE_PASS or MMD_CHUNKED == "0" or not (MMD_CHUNKED == "always" or d > 128 or d % 32):
        return None
    import ctypes
    out = (ctypes.c_int64 * 8)()
    if _lib.lib().gda_mmd_chunked_plan(int(times), int(n), int(d), float(kernel_mul), int(kernel_num), out, 8) != 0:
        return None
    return int(out[0]), int(out[1])


def mmd_one_pass_segments(times, n, d, kernel_mul=2.0, kernel_num=5):
    """Row segments of the one-pass MMD for these shapes; 0 when it does not cover them (or is switched off)."""
    if not MMD_ONE_PASS:
        return 0
    plan = mmd_chunked_plan(times, n, d, kernel_mul, kernel_num)
    if plan is not None:
        return plan[0]
    return int(_lib.lib().gda_mmd_fused_nseg(int(times), int(n), int(d), float(kernel_mul), int(kernel_num)))


_one_pass_checked = {}      # device index -> True (agrees with the two-pass kernels) | False (declined) | "running"
MMD_ONE_PASS_SELF_CHECK = _os.environ.get("PYGDA_AMD_MMD_SELF_CHECK", "1") == "1"


def mmd_one_pass_ok(dev):
    """First use of the one-pass MMD on a device: run it BESIDE the fp32-MFMA two-pass kernels on a small fixed problem
    (own generator: the global CPU stream that seeded runs depend on is not touched) and keep it only if loss and row
    gradients agree to 1e-4 -- the kernel is hand-scheduled gfx950 assembly (LDS-DMA, explicit waits) that only the GPU
    tests execute, so a library built against another driver / compiler is caught here, once, instead of training on a
    silently different loss.  ~10 launches, once per process and device; never inside a capture."""
    global MMD_ONE_PASS
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    state = _one_pass_checked.get(key)
    if state is not None:
        return state is not False
    if not MMD_ONE_PASS_SELF_CHECK:
        _one_pass_checked[key] = True
        return True
    if torch.cuda.is_current_stream_capturing():
        return True                                    # cannot compare inside a capture; checked at the next eager call
    _one_pass_checked[key] = "running"
    gen = torch.Generator().manual_seed(20240521)
    ok, why = True, ""
    try:
        for d in (128, 64, 200):                       # 200: the chunked kernel (two chunks of 128 | 96)
            times, n = 2, 80
            a = (torch.randn(times * n, d, generator=gen) * 0.7 + 0.3).to(dev).requires_grad_(True)
            b = (torch.randn(times * n, d, generator=gen) * 0.9 - 0.1).to(dev).requires_grad_(True)
            got = []
            for one in (True, False):
                MMD_ONE_PASS = one
                with torch.enable_grad():              # called from inside an autograd Function's forward
                    loss = _MMD.apply(a, b, None, None, times, n, 2.0, 5, None)
                    ga, gb = torch.autograd.grad(loss, (a, b))
                got.append((loss.detach().double(), torch.cat([ga, gb]).double()))
            MMD_ONE_PASS = True
            (l1, g1), (l2, g2) = got
            dl = float((l1 - l2).abs() / l2.abs().clamp_min(1e-30))
            dg = float((g1 - g2).norm() / g2.norm().clamp_min(1e-30))
            if not (dl <= 1e-4 and dg <= 1e-4):        # NaN fails too
                ok, why = False, f"d={d}: loss differs by {dl:.3g}, row gradients by {dg:.3g} (relative)"
                break
    except Exception as exc:                           # noqa: BLE001 -- a launch failure is a reason to decline too
        ok, why = False, f"{type(exc).__name__}: {exc}"
    finally:
        MMD_ONE_PASS = True
    _one_pass_checked[key] = ok
    if not ok:
        import warnings
        warnings.warn("one-pass MMD kernel declined by its first-use self-check (" + why +
                      "); this process uses the two-pass fp32-MFMA kernels")
    return ok


class _MMD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, tgt, src_idx, tgt_idx, times, n, kernel_mul, kernel_num, fix_sigma, sel=None, scale=1.0,
                add=None):
        """``(add +) scale * MMD``: the trainer's loss line ``loss = CE + MMD(...) * weight`` (a2gnn.py:207-209)
        without elementwise glue kernels around the loss kernels."""
        ctx.sel, ctx.scale, ctx.has_add = sel, float(scale), add is not None
        ctx.sinks = (sink_of(src), sink_of(tgt))
        src, tgt = _f32c(src, "source_feat"), _f32c(tgt, "target_feat")
        d = src.size(1)
        if tgt.size(1) != d:
            raise ValueError("source and target features must have the same width")
        dev = src.device
        m = 2 * n
        ctx.feat_rows = (src.size(0), tgt.size(0))
        idx_s = idx_t = rows_s = rows_t = None
        plan = mmd_chunked_plan(times, n, d, kernel_mul, kernel_num) if not fix_sigma else None
        if plan is not None and not mmd_one_pass_ok(dev):
            plan = None
        if src_idx is not None:
            idx_s, idx_t = src_idx.contiguous(), tgt_idx.contiguous()
            if plan is None and not (MMD_INDEX_IN_KERNEL and sel is not None):
                # the sampled rows gathered ONCE into [times*n, d] (2 x 2.5 MB at the A2GNN shapes) -- by the
                # statistics kernel, which reads them through the index anyway; everything after reads the copy
                rows_s = torch.empty(times * n, d, dtype=torch.float32, device=dev)
                rows_t = torch.empty(times * n, d, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        bw = torch.empty(times, dtype=torch.float32, device=dev)
        L = _lib.lib()
        addc = None if add is None else add.detach().to(torch.float32).reshape(1).contiguous()
        nseg = 0
        if plan is not None:
            nseg, dp = plan
            rows_s = torch.empty(times * n, dp, dtype=torch.float32, device=dev)     # gathered / copied, zero padded
            rows_t = torch.empty(times * n, dp, dtype=torch.float32, device=dev)
            part = torch.empty(times, nseg, m, dp, dtype=torch.float32, device=dev)
            ws = _lib.workspace(L.gda_mmd_chunked_workspace_bytes(times, n, d), dev, "mmd_chunked")
            with profiler.region("mmd_fwd", 5, 0, times * 3 * 4 * m * m * dp):
                _lib.check(L.gda_mmd_chunked_fwd_f32(
                    _lib.ptr(src), d, _lib.ptr(tgt), d, d, _lib.ptr(idx_s), _lib.ptr(idx_t), times, n, float(kernel_mul),
                    int(kernel_num), 0.0, float(scale), _lib.ptr(addc), _lib.ptr(rows_s), _lib.ptr(rows_t), dp,
                    _lib.ptr(loss), _lib.ptr(bw), _lib.ptr(part), dp, nseg, _lib.ptr(ws), ws.numel(), _lib.stream()),
                    "gda_mmd_chunked_fwd_f32")
            ctx.one_pass, ctx.ldp = nseg, dp
            ctx.save_for_backward(src_idx, tgt_idx, part)
            ctx.cfg = (times, n, d, float(kernel_mul), int(kernel_num))
            return loss.reshape(())
        ws = _lib.workspace(L.gda_mmd_workspace_bytes(times, n, d), dev, "mmd")
        if (idx_s is None or rows_s is not None) and not fix_sigma:
            nseg = mmd_one_pass_segments(times, n, d, kernel_mul, kernel_num)
            aligned = src.data_ptr() % 16 == 0 and tgt.data_ptr() % 16 == 0
            nseg = nseg if aligned and (not nseg or mmd_one_pass_ok(dev)) else 0
        ctx.one_pass, ctx.ldp = nseg, d
        if nseg:
            part = torch.empty(times, nseg, m, d, dtype=torch.float32, device=dev)
            with profiler.region("mmd_fwd", 4, 0, times * 3 * 4 * m * m * d):
                _lib.check(L.gda_mmd_fused_fwd_f32(
                    _lib.ptr(src), d, _lib.ptr(tgt), d, d, _lib.ptr(idx_s), _lib.ptr(idx_t), times, n, float(kernel_mul),
                    int(kernel_num), float(fix_sigma) if fix_sigma else 0.0, float(scale), _lib.ptr(addc),
                    _lib.ptr(rows_s), _lib.ptr(rows_t), _lib.ptr(loss), _lib.ptr(bw), _lib.ptr(part), nseg,
                    _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_mmd_fused_fwd_f32")
            ctx.save_for_backward(src_idx, tgt_idx, part)
            ctx.cfg = (times, n, d, float(kernel_mul), int(kernel_num))
            return loss.reshape(())
        l2 = torch.empty(times, m, m, dtype=torch.float32, device=dev)
        with profiler.region("mmd_fwd", 4, 0, times * (3 * m * m * d // 2 + 12 * m * m)):
            _lib.check(L.gda_mmd_fwd_gather_f32(
                _lib.ptr(src), d, _lib.ptr(tgt), d, d, _lib.ptr(idx_s), _lib.ptr(idx_t), times, n, float(kernel_mul),
                int(kernel_num), float(fix_sigma) if fix_sigma else 0.0, float(scale), _lib.ptr(addc),
                _lib.ptr(rows_s), _lib.ptr(rows_t), _lib.ptr(loss), _lib.ptr(bw), _lib.ptr(l2), _lib.ptr(ws), ws.numel(),
                _lib.stream()), "gda_mmd_fwd_gather_f32")
        if rows_s is not None:
            src, tgt = rows_s, rows_t
        ctx.save_for_backward(src, tgt, src_idx, tgt_idx, bw, l2)
        ctx.in_kernel = idx_s is not None and rows_s is None
        ctx.cfg = (times, n, float(kernel_mul), int(kernel_num))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        if ctx.one_pass:
            return _MMD._backward_one_pass(ctx, gl)
        src, tgt, src_idx, tgt_idx, bw, l2 = ctx.saved_tensors
        times, n, kernel_mul, kernel_num = ctx.cfg
        d, dev, m = src.size(1), src.device, 2 * n
        glc = gl.reshape(1).to(torch.float32).contiguous()
        L = _lib.lib()
        ws = _lib.workspace(L.gda_mmd_workspace_bytes(times, n, d), dev, "mmd")
        idx_s, idx_t = (src_idx, tgt_idx) if ctx.in_kernel else (None, None)
        g_add = gl if ctx.has_add else None
        tail = (None,) * 9 + (g_add,)
        if src_idx is not None and ctx.sel is not None and MMD_SCATTER_FUSED:
            # row gradients summed straight onto the sampled feature rows (segment reduce + scatter in one kernel)
            s_rp, s_ci, t_rp, t_ci, _ones = ctx.sel
            gs = torch.empty(ctx.feat_rows[0], d, dtype=torch.float32, device=dev)
            gt = torch.empty(ctx.feat_rows[1], d, dtype=torch.float32, device=dev)
            with profiler.region("mmd_bwd", 2, 0, times * (3 * m * m * d + 12 * m * m)):
                _lib.check(L.gda_mmd_bwd_ex_f32(
                    _lib.ptr(src), d, _lib.ptr(tgt), d, d, _lib.ptr(idx_s), _lib.ptr(idx_t), times, n, kernel_mul,
                    kernel_num, _lib.ptr(bw), _lib.ptr(l2), _lib.ptr(glc), ctx.scale, None,
                    _lib.ptr(s_rp), _lib.ptr(s_ci), ctx.feat_rows[0], _lib.ptr(gs),
                    _lib.ptr(t_rp), _lib.ptr(t_ci), ctx.feat_rows[1], _lib.ptr(gt),
                    _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_mmd_bwd_ex_f32")
            return (gs if ctx.needs_input_grad[0] else None, gt if ctx.needs_input_grad[1] else None) + tail
        grad_rows = torch.empty(times, m, d, dtype=torch.float32, device=dev)
        with profiler.region("mmd_bwd", 2, 0, times * (3 * m * m * d + 12 * m * m)):
            _lib.check(L.gda_mmd_bwd_ex_f32(
                _lib.ptr(src), d, _lib.ptr(tgt), d, d, None, None, times, n, kernel_mul, kernel_num,
                _lib.ptr(bw), _lib.ptr(l2), _lib.ptr(glc), ctx.scale, _lib.ptr(grad_rows),
                None, None, 0, None, None, None, 0, None, _lib.ptr(ws), ws.numel(), _lib.stream()),
                "gda_mmd_bwd_ex_f32")
        if src_idx is None:                       # rows as given, stacked [times, n, d]
            gs, gt = grad_rows[:, :n].reshape(times * n, d), grad_rows[:, n:].reshape(times * n, d)
        elif ctx.sel is not None:                 # selection CSRs prepared on the host with the samples
            s_rp, s_ci, t_rp, t_ci, ones = ctx.sel
            flat = grad_rows.view(times * m, d)
            gs = _selection_spmm(s_rp, s_ci, ones, flat, ctx.feat_rows[0])
            gt = _selection_spmm(t_rp, t_ci, ones, flat, ctx.feat_rows[1])
        else:
            gs = _scatter_rows(grad_rows, src_idx, 0, n, ctx.feat_rows[0])
            gt = _scatter_rows(grad_rows, tgt_idx, n, n, ctx.feat_rows[1])
        return (gs if ctx.needs_input_grad[0] else None, gt if ctx.needs_input_grad[1] else None) + tail


def _mmd_backward_one_pass(ctx, gl):
    """Backward of the one-pass MMD: the forward left unscaled row-gradient partials; this folds the segments, applies
    4 * dloss * scale / (n^2 times) and scatters onto the feature rows (one launch with the selection CSRs)."""
    src_idx, tgt_idx, part = ctx.saved_tensors
    times, n, d, _, _ = ctx.cfg
    dev, m = part.device, 2 * n
    glc = gl.reshape(1).to(torch.float32).contiguous()
    L = _lib.lib()
    g_add = gl if ctx.has_add else None
    tail = (None,) * 9 + (g_add,)
    if src_idx is not None and ctx.sel is not None and MMD_SCATTER_FUSED:
        s_rp, s_ci, t_rp, t_ci, _ones = ctx.sel
        # a feature matrix that came straight out of dropout(relu(.)) may have handed over a GradSink: its row gradients
        # are then stored already through that activation's backward (mask in the scatter's epilogue)
        ok = lambda k, rows: (k is not None and k.buf.shape == (rows, d) and k.buf.data_ptr() % 16 == 0
                              and k.y.data_ptr() % 16 == 0)
        ks = ctx.sinks[0] if ok(ctx.sinks[0], ctx.feat_rows[0]) and ctx.needs_input_grad[0] else None
        kt = ctx.sinks[1] if ok(ctx.sinks[1], ctx.feat_rows[1]) and ctx.needs_input_grad[1] else None
        gs = ks.buf if ks is not None else torch.empty(ctx.feat_rows[0], d, dtype=torch.float32, device=dev)
        gt = kt.buf if kt is not None else torch.empty(ctx.feat_rows[1], d, dtype=torch.float32, device=dev)
        with profiler.region("mmd_bwd", 1, 0, 0):
            _lib.check(L.gda_mmd_fused_bwd_ld_f32(
                _lib.ptr(part), ctx.ldp, ctx.one_pass, times, n, d, _lib.ptr(glc), ctx.scale, None,
                _lib.ptr(s_rp), _lib.ptr(s_ci), ctx.feat_rows[0], _lib.ptr(gs),
                _lib.ptr(t_rp), _lib.ptr(t_ci), ctx.feat_rows[1], _lib.ptr(gt),
                _lib.ptr(ks.y if ks is not None else None), ks.p if ks is not None else 0.0,
                _lib.ptr(kt.y if kt is not None else None), kt.p if kt is not None else 0.0, _lib.stream()),
                "gda_mmd_fused_bwd_ld_f32")
        for k in (ks, kt):
            if k is not None:
                k.written = True
        return (gs if ctx.needs_input_grad[0] else None, gt if ctx.needs_input_grad[1] else None) + tail
    grad_rows = torch.empty(times, m, d, dtype=torch.float32, device=dev)
    with profiler.region("mmd_bwd", 1, 0, 0):
        _lib.check(L.gda_mmd_fused_bwd_ld_f32(
            _lib.ptr(part), ctx.ldp, ctx.one_pass, times, n, d, _lib.ptr(glc), ctx.scale, _lib.ptr(grad_rows),
            None, None, 0, None, None, None, 0, None, None, 0.0, None, 0.0, _lib.stream()), "gda_mmd_fused_bwd_ld_f32")
    if src_idx is None:                       # rows as given, stacked [times, n, d]
        gs, gt = grad_rows[:, :n].reshape(times * n, d), grad_rows[:, n:].reshape(times * n, d)
    elif ctx.sel is not None:
        s_rp, s_ci, t_rp, t_ci, ones = ctx.sel
        flat = grad_rows.view(times * m, d)
        gs = _selection_spmm(s_rp, s_ci, ones, flat, ctx.feat_rows[0])
        gt = _selection_spmm(t_rp, t_ci, ones, flat, ctx.feat_rows[1])
    else:
        gs = _scatter_rows(grad_rows, src_idx, 0, n, ctx.feat_rows[0])
        gt = _scatter_rows(grad_rows, tgt_idx, n, n, ctx.feat_rows[1])
    return (gs if ctx.needs_input_grad[0] else None, gt if ctx.needs_input_grad[1] else None) + tail


_MMD._backward_one_pass = staticmethod(_mmd_backward_one_pass)


class _PinnedRing:
    """A few pinned host blocks handed out in turn, each guarded by the event of the copy that last left it: what a
    step ships to the device (MMD row samples + their selection CSRs, ~1.4 MB at cfg-S) goes as ONE asynchronous
    copy.  ``tensor.to(device)`` from pageable memory is synchronous -- the host waits for the stream to drain
    first -- which serialised the eager sampled-training loop against the GPU (six such copies per step: 1.2 ms)."""

    def __init__(self, slots=4):
        self.slots, self.turn = [None] * slots, 0

    def take(self, nbytes):
        i = self.turn
        self.turn = (i + 1) % len(self.slots)
        hit = self.slots[i]
        if hit is not None and hit[1] is not None:
            hit[1].synchronize()                       # the copy out of this block (several steps ago) has finished
        if hit is None or hit[0].numel() < nbytes:
            cap = 1 << 16
            while cap < nbytes + nbytes // 2:          # headroom: batches differ in size, pinning is expensive (~50 ms)
                cap <<= 1
            hit = [torch.empty(cap, dtype=torch.uint8).pin_memory(), None]
            self.slots[i] = hit
        return i, hit[0]

    def sent(self, i, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.slots[i][1] = ev


_pinned_ring = _PinnedRing()


def mmd_samples_to_device(source_sample, target_sample, ns, nt, dev, stacked=True):
    """The two ``[times, n]`` CPU row-sample tensors of an MMD call and the selection CSRs of their gradient scatter
    -> ``(idx_s, idx_t, sel)`` on the device through one pinned block and one non-blocking copy.  ``stacked``: the
    gradient buffer holds source and target rows side by side (``[times, 2n, d]``, the single-process loss);
    otherwise each domain scatters out of its own ``[times, n, d]`` buffer (the data-parallel row exchange)."""
    times, n = source_sample.shape
    m = 2 * n if stacked else n
    a16 = lambda v: (v + 15) // 16 * 16
    sizes = [8 * times * n, 8 * times * n, 4 * (ns + 1), 4 * times * n, 4 * (nt + 1), 4 * times * n]
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + a16(sz))
    slot, host = _pinned_ring.take(offs[-1])
    view = lambda k, dt, cnt: host[offs[k]:offs[k] + sizes[k]].view(dt)[:cnt]
    view(0, torch.int64, times * n).copy_(source_sample.reshape(-1))
    view(1, torch.int64, times * n).copy_(target_sample.reshape(-1))
    selection_csr_host(source_sample, ns, 0, m, out=(view(2, torch.int32, ns + 1), view(3, torch.int32, times * n)))
    selection_csr_host(target_sample, nt, n if stacked else 0, m,
                       out=(view(4, torch.int32, nt + 1), view(5, torch.int32, times * n)))
    block = torch.empty(offs[-1], dtype=torch.uint8, device=dev)
    from .hipgraph import _ship
    _ship(block, host[:offs[-1]])              # a kernel of this stream that reads the pinned block (no DMA-engine hand-over)
    _pinned_ring.sent(slot, torch.cuda.current_stream())
    dview = lambda k, dt, cnt: block[offs[k]:offs[k] + sizes[k]].view(dt)[:cnt]
    sel = (dview(2, torch.int32, ns + 1), dview(3, torch.int32, times * n),
           dview(4, torch.int32, nt + 1), dview(5, torch.int32, times * n),
           torch.ones(times * n, dtype=torch.float32, device=dev))
    return dview(0, torch.int64, times * n).view(times, n), dview(1, torch.int64, times * n).view(times, n), sel


def selection_csr_host(idx, num_feat_rows, offset, m, out=None):
    """Host side of the sample-gradient scatter (native counting sort, csrc/gda_sampler.cpp):
    for ``idx [times, n]`` (CPU int64 row samples) the CSR of the 0/1 selection matrix -- rows =
    feature rows, columns = positions ``t*m + offset + r`` of the ``[times, m, d]`` gradient
    buffer.  ``out = (rowptr, colidx)`` may name preallocated (pinned) CPU int32 tensors."""
    idx = idx.contiguous()
    times, n = idx.shape
    if out is None:
        out = (torch.empty(num_feat_rows + 1, dtype=torch.int32), torch.empty(times * n, dtype=torch.int32))
    L = _lib.lib()
    _lib.check(L.gda_selection_csr_host(idx.data_ptr(), int(times), int(n), int(num_feat_rows), int(offset),
                                        int(m), out[0].data_ptr(), out[1].data_ptr()),
               "gda_selection_csr_host")
    return out


def _selection_spmm(rowptr, colidx, ones, flat, num_feat_rows):
    d = flat.size(1)
    out = torch.empty(num_feat_rows, d, dtype=torch.float32, device=flat.device)
    L = _lib.lib()
    _lib.check(L.gda_spmm_csr_f32(_lib.ptr(rowptr), _lib.ptr(colidx), _lib.ptr(ones), num_feat_rows, d,
                                  _lib.ptr(flat), d, _lib.ptr(out), d, None, _lib.stream()),
               "gda_spmm_csr_f32")
    return out


def _scatter_rows(grad_rows, idx, offset, n, num_feat_rows):
    """Sum the per-sample row gradients back onto the sampled feature rows (duplicates
    included) deterministically: a CSR SpMM with the 0/1 selection matrix, built on device
    by the same ingestion kernel (no self loops, no normalisation)."""
    times, m, d = grad_rows.shape
    dev = grad_rows.device
    pos = (torch.arange(times, device=dev).view(-1, 1) * m + offset +
           torch.arange(n, device=dev).view(1, -1)).reshape(-1)
    n_nodes = max(num_feat_rows, times * m)
    g = build_csr(torch.stack([pos, idx.reshape(-1)]), n_nodes, None, add_self_loops=False,
                  normalize=False, validate=False)
    flat = grad_rows.view(times * m, d)
    if n_nodes > times * m:                       # SpMM gathers rows < n_nodes only via colidx: fine
        pass
    out = torch.empty(num_feat_rows, d, dtype=torch.float32, device=dev)
    L = _lib.lib()
    _lib.check(L.gda_spmm_csr_f32(_lib.ptr(g.rowptr), _lib.ptr(g.colidx), _lib.ptr(g.val),
                                  num_feat_rows, d, _lib.ptr(flat), d, _lib.ptr(out), d, None,
                                  _lib.stream()), "gda_spmm_csr_f32")
    return out


class _SampleRows(torch.autograd.Function):
    """``feat[idx]`` for ``idx [times, n]`` -> ``[times, n, d]``: gather kernel forward,
    deterministic selection-matrix SpMM backward (duplicates summed in a fixed order)."""

    @staticmethod
    def forward(ctx, feat, idx, sel=None):
        ctx.save_for_backward(idx)
        ctx.rows, ctx.sel = feat.size(0), sel
        return gather_rows(feat, idx.reshape(-1)).view(idx.size(0), idx.size(1), feat.size(1))

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        times, n, d = g.shape
        g = g.contiguous()
        if ctx.sel is not None:               # selection CSR prepared on the host with the draws
            rowptr, colidx, ones = ctx.sel
            return _selection_spmm(rowptr, colidx, ones, g.view(times * n, d), ctx.rows), None, None
        return _scatter_rows(g.view(times, n, d), idx, 0, n, ctx.rows), None, None


def sample_rows(feat, idx, sel=None):
    """``sel = (rowptr, colidx, ones)``: device CSR of the 0/1 selection matrix of ``idx`` (columns
    ``t * n + r``; :func:`selection_csr_host` with ``offset=0, m=n``) -- without it the backward
    pass builds that CSR on the device (a sort per call)."""
    return _SampleRows.apply(_f32c(feat, "feat"), idx, sel)


def mmd_loss_rows(source_rows, target_rows, kernel_mul=2.0, kernel_num=5, fix_sigma=None):
    """MMD over explicit row sets ``[times, n, d]`` per domain (the data-parallel path, where
    the rows are an all-gather of every rank's samples)."""
    if source_rows.shape != target_rows.shape or source_rows.dim() != 3:
        raise RuntimeError("row sets must both be [times, n, d]")
    times, n, d = source_rows.shape
    return _MMD.apply(source_rows.reshape(times * n, d), target_rows.reshape(times * n, d), None, None,
                      int(times), int(n), kernel_mul, kernel_num, fix_sigma)


def mmd_loss(source_feat, target_feat, src_idx=None, tgt_idx=None, kernel_mul=2.0, kernel_num=5,
             fix_sigma=None, sel=None, scale=1.0, add=None):
    """Sampled multi-kernel MMD (mmd.py:57-159).  ``src_idx/tgt_idx``: ``[times, n]`` int64
    device tensors of row samples, or both ``None`` for get_MMD on the rows as given."""
    if (src_idx is None) != (tgt_idx is None):
        raise ValueError("give both index tensors or neither")
    if src_idx is None:
        if source_feat.size(0) != target_feat.size(0):
            # the reference's XX + YY - XY - YX broadcast fails the same way (mmd.py:100-106)
            raise RuntimeError("get_MMD needs equally many source and target rows, got "
                               f"{source_feat.size(0)} and {target_feat.size(0)}")
        times, n = 1, source_feat.size(0)
    else:
        _lib.require_gpu_tensor(src_idx, "src_idx", torch.int64)
        _lib.require_gpu_tensor(tgt_idx, "tgt_idx", torch.int64)
        if src_idx.shape != tgt_idx.shape or src_idx.dim() != 2:
            raise RuntimeError("source and target samples must both be [times, sampling_num]")
        times, n = src_idx.shape
        src_idx, tgt_idx = src_idx.contiguous(), tgt_idx.contiguous()
    return _MMD.apply(source_feat, target_feat, src_idx, tgt_idx, int(times), int(n), kernel_mul,
                      kernel_num, fix_sigma, sel, scale, add)


# ------------------------------------------------- GRL + discriminator + CE (fused) --
class _GrlDiscCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fs, ft, W, b, alpha, labels):
        fs, ft, W, b = _f32c(fs, "source_feat"), _f32c(ft, "target_feat"), _f32c(W, "weight"), _f32c(b, "bias")
        ns, nt, h, C = fs.size(0), ft.size(0), fs.size(1), W.size(0)
        dev = fs.device
        probs = torch.empty(ns + nt, C, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = _lib.workspace(L.gda_grl_disc_workspace_bytes(ns + nt, h, C), dev, "disc")
        _lib.check(L.gda_grl_disc_ce_fwd_f32(
            _lib.ptr(fs), h, ns, _lib.ptr(ft), h, nt, h, C, _lib.ptr(W), _lib.ptr(b), _lib.ptr(labels),
            _lib.ptr(probs), _lib.ptr(loss), _lib.ptr(ws), ws.numel(), _lib.stream()),
            "gda_grl_disc_ce_fwd_f32")
        ctx.save_for_backward(fs, ft, W, probs, labels)
        ctx.alpha = float(alpha)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        fs, ft, W, probs, labels = ctx.saved_tensors
        ns, nt, h, C = fs.size(0), ft.size(0), fs.size(1), W.size(0)
        dev = fs.device
        gfs = torch.empty_like(fs) if ctx.needs_input_grad[0] else None
        gft = torch.empty_like(ft) if ctx.needs_input_grad[1] else None
        gW, gb = torch.empty_like(W), torch.empty(C, dtype=torch.float32, device=dev)
        gl = gl.reshape(1).to(torch.float32).contiguous()
        L = _lib.lib()
        ws = _lib.workspace(L.gda_grl_disc_workspace_bytes(ns + nt, h, C), dev, "disc")
        _lib.check(L.gda_grl_disc_ce_bwd_f32(
            _lib.ptr(fs), h, ns, _lib.ptr(ft), h, nt, h, C, _lib.ptr(W), _lib.ptr(labels),
            _lib.ptr(probs), _lib.ptr(gl), ctx.alpha, _lib.ptr(gfs), _lib.ptr(gft), _lib.ptr(gW),
            _lib.ptr(gb), _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_grl_disc_ce_bwd_f32")
        return gfs, gft, gW, gb, None, None


def grl_disc_ce(source_feat, target_feat, weight, bias, alpha, labels=None):
    """``F.cross_entropy(Linear(GradReverse(cat(source, target))), domain_labels)`` fused
    (a2gnn.py:197-205 / grade.py:170-176).  Default labels: 0 for source rows, 1 for target.
    ``alpha`` may be a 0-dim DEVICE tensor (the captured step updates it per epoch without touching
    the graph): the fused kernel then runs with the reversal folded out (alpha = -1) behind a
    tensor-valued GradReverse."""
    if torch.is_tensor(alpha):
        from .nn.reverse_layer import GradReverse
        return _GrlDiscCE.apply(GradReverse.apply(source_feat, alpha), GradReverse.apply(target_feat, alpha),
                                weight, bias, -1.0, labels)
    return _GrlDiscCE.apply(source_feat, target_feat, weight, bias, alpha, labels)


# ------------------------------ GRL + two-layer discriminator + per-domain CE (fused) --
def _grl_mlp_fwd(ctx, es, et, W1, b1, W2, b2, alpha, p, head=0):
    ctx.head = head
    es, et = _f32c(es, "source_feat"), _f32c(et, "target_feat")
    W1, b1, W2, b2 = _f32c(W1, "W1"), _f32c(b1, "b1"), _f32c(W2, "W2"), _f32c(b2, "b2")
    ns, nt, h, a = es.size(0), et.size(0), es.size(1), W1.size(0)
    dev = es.device
    st = dropout_state
    if st.seed is None:
        st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
    site = st.next_site()
    st.next_site()                                   # two call sites: source rows, target rows
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    L = _lib.lib()
    ws = _lib.workspace(L.gda_grl_mlp_ce_workspace_bytes(h, a), dev, "disc_mlp")
    with profiler.region(f"disc_mlp_fwd[{h}x{a}]", 2, 4 * (ns + nt) * h, 2 * (ns + nt) * a * (h + W2.size(0))):
        _lib.check(L.gda_mlp_head_fwd_f32(
            head, _lib.ptr(es), h, ns, _lib.ptr(et), h, nt, h, a, _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(W2), _lib.ptr(b2),
            float(p), ctypes.c_uint64(st.seed), _lib.ptr(st.counter(dev)), ctypes.c_uint32(site),
            _lib.ptr(losses), _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_mlp_head_fwd_f32")
    ctx.save_for_backward(es, et, W1, b1, W2, b2, alpha if torch.is_tensor(alpha) else None)
    ctx.alpha = None if torch.is_tensor(alpha) else float(alpha)
    ctx.p, ctx.seed, ctx.site = float(p), st.seed, site
    return losses


def _grl_mlp_bwd(ctx, g, stride):
    es, et, W1, b1, W2, b2, alpha_dev = ctx.saved_tensors
    ns, nt, h, a = es.size(0), et.size(0), es.size(1), W1.size(0)
    dev = es.device
    ges = torch.empty_like(es) if ctx.needs_input_grad[0] else None
    get = torch.empty_like(et) if ctx.needs_input_grad[1] else None
    gW1, gb1, gW2, gb2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2), torch.empty_like(b2)
    if alpha_dev is not None:
        alpha_dev = alpha_dev.detach().reshape(1).to(torch.float32)
    L = _lib.lib()
    ws = _lib.workspace(L.gda_grl_mlp_ce_workspace_bytes(h, a), dev, "disc_mlp")
    # recomputed first layer, d hidden . W1 (input gradients), d hidden^T x (weight gradient): 3 products of rows x a x h
    with profiler.region(f"disc_mlp_bwd[{h}x{a}]", 2, 4 * (ns + nt) * h * (2 if ges is not None else 1), 6 * (ns + nt) * a * h):
        _lib.check(L.gda_mlp_head_bwd_f32(
            ctx.head, _lib.ptr(es), h, ns, _lib.ptr(et), h, nt, h, a, _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(W2), _lib.ptr(b2),
            ctx.p, ctypes.c_uint64(ctx.seed), _lib.ptr(dropout_state.counter(dev)), ctypes.c_uint32(ctx.site),
            _lib.ptr(g), stride, 0.0 if ctx.alpha is None else ctx.alpha, _lib.ptr(alpha_dev),
            _lib.ptr(ges), _lib.ptr(get), _lib.ptr(gW1), _lib.ptr(gb1), _lib.ptr(gW2), _lib.ptr(gb2),
            _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_mlp_head_bwd_f32")
    return ges, get, gW1, gb1, gW2, gb2, None, None


class _GrlMlpCESum(torch.autograd.Function):
    """The two domain means added, as the reference adds them: one scalar out, one scalar gradient in."""

    @staticmethod
    def forward(ctx, es, et, W1, b1, W2, b2, alpha, p):
        return _grl_mlp_fwd(ctx, es, et, W1, b1, W2, b2, alpha, p)[2]

    @staticmethod
    def backward(ctx, g):
        return _grl_mlp_bwd(ctx, g.reshape(1).to(torch.float32).contiguous(), 0)


class _GrlMlpCEPair(torch.autograd.Function):
    """The two domain means separately (data-parallel callers weight them by node counts)."""

    @staticmethod
    def forward(ctx, es, et, W1, b1, W2, b2, alpha, p):
        losses = _grl_mlp_fwd(ctx, es, et, W1, b1, W2, b2, alpha, p)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_s, g_t):
        return _grl_mlp_bwd(ctx, torch.stack([g_s.reshape(()).to(torch.float32), g_t.reshape(()).to(torch.float32)]), 1)


class _CriticMeans(torch.autograd.Function):
    """``(mean D(source rows), mean D(target rows))`` for ``D = Linear(h, a) - ReLU - Dropout(p) - Linear(a, 1) - Sigmoid``
    (AdaGCN's critic inside the ENCODER's loss, adagcn.py:190-193): the row kernels of the two-layer discriminator with
    the sigmoid-mean head (gda_mlp_head_*_f32, head = 1), no gradient reversal."""

    @staticmethod
    def forward(ctx, es, et, W1, b1, W2, b2, p):
        losses = _grl_mlp_fwd(ctx, es, et, W1, b1, W2, b2, -1.0, p, head=1)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_s, g_t):
        return _grl_mlp_bwd(ctx, torch.stack([g_s.reshape(()).to(torch.float32), g_t.reshape(()).to(torch.float32)]), 1)[:7]


class _CriticMeansVec(torch.autograd.Function):
    """:class:`_CriticMeans` with the kernel's result block ``[mean D(source), mean D(target), unused]`` handed on as ONE
    tensor: its gradient arrives as one tensor too (no stack in front of the backward kernels)."""

    @staticmethod
    def forward(ctx, es, et, W1, b1, W2, b2, p):
        return _grl_mlp_fwd(ctx, es, et, W1, b1, W2, b2, -1.0, p, head=1)

    @staticmethod
    def backward(ctx, g):
        return _grl_mlp_bwd(ctx, g.to(torch.float32).contiguous(), 1)[:7]


class _AddAbsGap(torch.autograd.Function):
    """``base + weight * |v[0] - v[1]|`` for a scalar ``base`` and a device vector ``v`` in four small launches forward and
    one backward (the composed expression and its autograd graph: seventeen)."""
    _wvec = {}

    @staticmethod
    def forward(ctx, base, v, weight):
        key = (float(weight), v.device, v.numel())
        wv = _AddAbsGap._wvec.get(key)
        if wv is None:
            wv = _AddAbsGap._wvec[key] = torch.tensor([weight, -weight] + [0.0] * (v.numel() - 2), dtype=torch.float32, device=v.device)
        d = v[0] - v[1]
        svec = torch.sign(d) * wv                         # d |gap| / d v, times the weight
        ctx.save_for_backward(svec)
        return torch.addcmul(base, d, svec[0])            # d * sign(d) * weight = weight * |d|

    @staticmethod
    def backward(ctx, g):
        (svec,) = ctx.saved_tensors
        return g, g * svec, None


def critic_abs_gap_loss(base, source_feat, target_feat, W1, b1, W2, b2, dropout_p, weight):
    """``base + weight * |mean D(source) - mean D(target)|`` (AdaGCN's encoder loss, adagcn.py:186-193) for the
    sigmoid-headed two-layer critic: the fused means (:func:`critic_means`) and the scalar tail in a handful of launches."""
    v = _CriticMeansVec.apply(source_feat, target_feat, W1, b1, W2, b2, dropout_p)
    return _AddAbsGap.apply(base, v, float(weight))


def critic_means_ok(es, W1, W2):
    return (es.is_cuda and es.dtype == torch.float32 and es.dim() == 2 and es.size(1) <= 128
            and W1.size(0) <= 64 and W2.dim() == 2 and W2.size(0) == 1)


def critic_means(source_feat, target_feat, W1, b1, W2, b2, dropout_p=0.0):
    """The two means ``torch.mean(D(source)), torch.mean(D(target))`` of a sigmoid-headed two-layer critic, fused."""
    return _CriticMeans.apply(source_feat, target_feat, W1, b1, W2, b2, dropout_p)


def grl_mlp_ce_ok(es, W1, W2):
    """Shapes the fused two-layer discriminator covers (csrc/gda_disc_mlp.hip)."""
    return (es.is_cuda and es.dtype == torch.float32 and es.dim() == 2 and es.size(1) <= 128
            and W1.size(0) <= 64 and W2.size(0) == 2)


def grl_mlp_ce(source_feat, target_feat, W1, b1, W2, b2, alpha, dropout_p=0.0, pair=False):
    """``CE(D(GradReverse(source)), 0).mean() + CE(D(GradReverse(target)), 1).mean()`` (``pair``: the two means as a
    tuple) for the two-layer discriminator ``D = Linear(h, a) - ReLU - Dropout(p) - Linear(a, 2)`` of UDAGCN
    (udagcn.py:176-190, udagcn_base.py:157-162), fused: one row kernel and one fold each way (include/gda_hip.h:
    gda_grl_mlp_ce_fwd_f32 / _bwd_f32).  ``alpha`` may be a 0-dim device tensor (captured steps)."""
    fn = _GrlMlpCEPair if pair else _GrlMlpCESum
    return fn.apply(source_feat, target_feat, W1, b1, W2, b2, alpha, dropout_p)


# ------------------------------------------------------ LSGAN discriminator head (DANE) --
class _LsganHead(torch.autograd.Function):
    """``mean((Linear(h, 1)(relu(x W1^T + b1)) - target) ** 2)``: first layer on the matrix-core kernels (bias in the
    epilogue), everything after it in one pass each way over Z (gda_lsgan_head_fwd/bwd_f32); gW1 / gb1 / gx from gZ
    by the TN (column sums as a by-product) and NN kernels."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, target):
        x, W1, b1 = _f32c(x, "rows"), _f32c(W1, "W1"), _f32c(b1, "b1")
        w2, b2 = _f32c(W2, "W2").reshape(-1), _f32c(b2, "b2").reshape(-1)
        rows, a = x.size(0), W1.size(0)
        Z = gemm(GEMM_NT, x, W1, bias=b1)
        pre = torch.empty(rows, dtype=torch.float32, device=x.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws = _lib.workspace(L.gda_lsgan_head_workspace_bytes(a), x.device, "lsgan")
        _lib.check(L.gda_lsgan_head_fwd_f32(_lib.ptr(Z), a, rows, a, _lib.ptr(w2), _lib.ptr(b2), float(target), _lib.ptr(pre),
                                            _lib.ptr(loss), _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_lsgan_head_fwd_f32")
        ctx.save_for_backward(x, W1, Z, w2, pre)
        ctx.target, ctx.w2_shape, ctx.b2_shape = float(target), W2.shape, b2.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x, W1, Z, w2, pre = ctx.saved_tensors
        rows, a = Z.shape
        gZ = torch.empty_like(Z)
        gw2 = torch.empty(a, dtype=torch.float32, device=Z.device)
        gb2 = torch.empty(1, dtype=torch.float32, device=Z.device)
        g = g.reshape(1).to(torch.float32).contiguous()
        L = _lib.lib()
        ws = _lib.workspace(L.gda_lsgan_head_workspace_bytes(a), Z.device, "lsgan")
        _lib.check(L.gda_lsgan_head_bwd_f32(_lib.ptr(Z), a, rows, a, _lib.ptr(w2), _lib.ptr(pre), ctx.target, _lib.ptr(g),
                                            _lib.ptr(gZ), a, _lib.ptr(gw2), _lib.ptr(gb2), _lib.ptr(ws), ws.numel(),
                                            _lib.stream()), "gda_lsgan_head_bwd_f32")
        gx = gemm(GEMM_NN, gZ, W1) if ctx.needs_input_grad[0] else None
        gb1 = torch.empty(a, dtype=torch.float32, device=Z.device)
        gW1 = gemm(GEMM_TN, gZ, x, colsum=gb1)
        return gx, gW1, gb1, gw2.reshape(ctx.w2_shape), gb2.reshape(ctx.b2_shape), None


def lsgan_head_ok(x, W1, W2):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= x.size(0) <= 200_000
            and W1.size(0) <= 256 and W1.size(1) <= 256 and W2.size(0) == 1)


def lsgan_head(x, W1, b1, W2, b2, target):
    """``((D(x) - target) ** 2).mean()`` for DANE's ``D = Linear(h, a) - ReLU - Linear(a, 1)`` (dane.py:241-247,
    339-350, 468-470): one GEMM, one row pass and one fold forward; one row pass, one fold and the GEMMs of the first
    layer backward."""
    return _LsganHead.apply(x, W1, b1, W2, b2, target)


# ------------------------------------------- Wasserstein critic update (WGAN-GP), fused --
def wgan_critic_grads(es, et, idx_s, idx_t, alpha, W1, b1, W2, b2, dropout_p, gp_weight, out):
    """Loss and parameter gradients of one critic update of AdaGCN (pygda/models/adagcn.py:169-183,387-454)
    in closed form (include/gda_hip.h: gda_wgan_critic_f32): ``out = (loss [1], gW1, gb1, gW2, gb2)``
    preallocated like the parameters -- the gradients land where the critic's optimiser reads them."""
    es, et = _f32c(es, "encoded_source"), _f32c(et, "encoded_target")
    n_s, h = es.shape
    n_t = et.size(0)
    a = W1.size(0)
    n_i = 0 if idx_s is None else idx_s.numel()
    L = _lib.lib()
    ws = _lib.workspace(L.gda_wgan_critic_workspace_bytes(n_s, n_t, n_i, h, a), es.device, "critic")
    st = dropout_state
    if st.seed is None:
        st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
    site = st.next_site()
    st.next_site(); st.next_site()                          # three mask sets: D(es), D(et), penalty rows
    loss, gW1, gb1, gW2, gb2 = out
    # algorithmic work of one update: per penalty row the products W1 x, W1^T u, W1 y and the row's u y^T (2 a h flops
    # each), per gap row its u x^T; the encodings read once, two rows gathered per interpolate
    rows_gp = n_s + n_t + n_i
    with profiler.region(f"wgan_critic[{h}x{a}]", 2 if h in (64, 96, 128) and a % 4 == 0 else 7,
                         4 * h * (n_s + n_t + 2 * n_i), 2 * a * h * (4 * rows_gp + n_s + n_t)):
        _lib.check(L.gda_wgan_critic_f32(
            _lib.ptr(es), n_s, _lib.ptr(et), n_t, h, _lib.ptr(idx_s), _lib.ptr(idx_t),
            _lib.ptr(None if alpha is None else _f32c(alpha, "alpha")), n_i,
            _lib.ptr(_f32c(W1, "W1")), _lib.ptr(_f32c(b1, "b1")), _lib.ptr(_f32c(W2, "W2")), _lib.ptr(_f32c(b2, "b2")), a,
            float(dropout_p), ctypes.c_uint64(st.seed), _lib.ptr(st.counter(es.device)), ctypes.c_uint32(site),
            float(gp_weight), _lib.ptr(loss), _lib.ptr(gW1), _lib.ptr(gb1), _lib.ptr(gW2), _lib.ptr(gb2),
            _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_wgan_critic_f32")
    return loss


def wgan_critic_adam(es, et, idx_s, idx_t, alpha, params, optimizer, dropout_p, gp_weight, loss):
    """One iteration of AdaGCN's critic loop INCLUDING ``c_optimizer.step()`` (pygda/models/adagcn.py:169-183) in the two
    launches of :func:`wgan_critic_grads` (include/gda_hip.h: gda_wgan_critic_adam_f32): ``params = (W1, b1, W2, b2)`` with
    preallocated ``.grad``, ``optimizer`` the :class:`pygda_amd.optim.Adam` that holds exactly them in one group.  Returns
    False -- nothing launched, nothing changed -- when the shapes are not the matrix-core path's or the optimiser is not
    of that form; the caller then runs :func:`wgan_critic_grads` + ``optimizer.step()``: the same bits."""
    from .optim import Adam
    if not isinstance(optimizer, Adam) or len(optimizer.param_groups) != 1 or optimizer.grad_aliases:
        return False
    group = optimizer.param_groups[0]
    if len(group["params"]) != 4 or any(a is not b for a, b in zip(group["params"], params)):
        return False
    es, et = _f32c(es, "encoded_source"), _f32c(et, "encoded_target")
    n_s, h = es.shape
    n_t, a = et.size(0), params[0].size(0)
    W1 = params[0]
    if not (h in (64, 96, 128) and a % 4 == 0 and all(p.is_contiguous() and p.dtype == torch.float32 for p in params)
            and (es.data_ptr() | et.data_ptr() | W1.data_ptr()) % 16 == 0):
        return False
    n_i = 0 if idx_s is None else idx_s.numel()
    L = _lib.lib()
    ws = _lib.workspace(L.gda_wgan_critic_workspace_bytes(n_s, n_t, n_i, h, a), es.device, "critic")
    table = (_lib.AdamTensorStruct * 4)()
    for k, p in enumerate(params):
        stt = optimizer._state(p)
        if p.grad is None or not p.grad.is_contiguous() or any(stt[n].stride() != p.stride() for n in ("exp_avg", "exp_avg_sq")):
            return False
        table[k] = _lib.AdamTensorStruct(p.data_ptr(), p.grad.data_ptr(), stt["exp_avg"].data_ptr(),
                                         stt["exp_avg_sq"].data_ptr(), stt["step"].data_ptr(), p.numel())
    st = dropout_state
    if st.seed is None:
        st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
    site = st.next_site()
    st.next_site(); st.next_site()                          # three mask sets: D(es), D(et), penalty rows
    b1_, b2_ = group["betas"]
    rows_gp = n_s + n_t + n_i
    with profiler.region(f"wgan_critic[{h}x{a}]", 2, 4 * h * (n_s + n_t + 2 * n_i), 2 * a * h * (4 * rows_gp + n_s + n_t)):
        _lib.check(L.gda_wgan_critic_adam_f32(
            _lib.ptr(es), n_s, _lib.ptr(et), n_t, h, _lib.ptr(idx_s), _lib.ptr(idx_t),
            _lib.ptr(None if alpha is None else _f32c(alpha, "alpha")), n_i, a,
            float(dropout_p), ctypes.c_uint64(st.seed), _lib.ptr(st.counter(es.device)), ctypes.c_uint32(site),
            float(gp_weight), _lib.ptr(loss), table, float(group["lr"]), float(b1_), float(b2_), float(group["eps"]),
            float(group["weight_decay"]), _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_wgan_critic_adam_f32")
    return True


# -------------------------------------------------------------------------- gather --
def gather_rows(x, idx):
    """``x[idx]`` for a ``[N, d]`` fp32 feature matrix (mini-batch assembly)."""
    x = _f32c(x, "x")
    _lib.require_gpu_tensor(idx, "idx", torch.int64)
    idx = idx.contiguous()
    out = torch.empty(idx.numel(), x.size(1), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    _lib.check(L.gda_gather_rows_f32(_lib.ptr(x), x.size(1), x.size(1), _lib.ptr(idx), idx.numel(),
                                     _lib.ptr(out), x.size(1), _lib.stream()), "gda_gather_rows_f32")
    return out


# ---------------------------------------------------------- graph-level readout --
class _SegmentMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg_ptr, batch):
        x = _f32c(x, "x")
        G, d = seg_ptr.numel() - 1, x.size(1)
        out = torch.empty(G, d, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().gda_segment_mean_fwd_f32(_lib.ptr(x), d, _lib.ptr(seg_ptr), G, d, _lib.ptr(out), d,
                                                       _lib.stream()), "gda_segment_mean_fwd_f32")
        ctx.save_for_backward(seg_ptr, batch)
        ctx.n = x.size(0)
        return out

    @staticmethod
    def backward(ctx, gout):
        seg_ptr, batch = ctx.saved_tensors
        gout = gout.contiguous()
        d = gout.size(1)
        gx = torch.empty(ctx.n, d, dtype=torch.float32, device=gout.device)
        _lib.check(_lib.lib().gda_segment_mean_bwd_f32(_lib.ptr(gout), d, _lib.ptr(seg_ptr), _lib.ptr(batch), ctx.n, d,
                                                       _lib.ptr(gx), d, _lib.stream()), "gda_segment_mean_bwd_f32")
        return gx, None, None


def segment_ptr(batch, size=None):
    """``seg_ptr [G+1]`` of a SORTED graph-index vector (what a DataLoader's collation produces), cached on the
    tensor; an unsorted vector is rejected (one device check per vector): the readout kernel walks contiguous rows."""
    hit = getattr(batch, "_gda_seg_ptr", None)
    if hit is not None and (size is None or hit.numel() - 1 == size):
        return hit
    _lib.require_gpu_tensor(batch, "batch", torch.int64)
    if batch.numel() > 1 and not getattr(batch, "_gda_sorted", False) and bool((batch[1:] < batch[:-1]).any()):
        raise ValueError("global_mean_pool: `batch` must be sorted (nodes of a graph contiguous), as PyG's DataLoader "
                         "collates it")
    G = (int(batch[-1]) + 1 if batch.numel() else 0) if size is None else int(size)
    seg = torch.zeros(G + 1, dtype=torch.int64, device=batch.device)
    if batch.numel():
        seg[1:] = torch.cumsum(torch.bincount(batch, minlength=G), 0)
    batch._gda_seg_ptr = seg
    return seg


def segment_mean(x, batch, size=None):
    """``global_mean_pool(x, batch)`` (pygda/nn/a2gnn_base.py:141) for a sorted ``batch``."""
    return _SegmentMean.apply(x, segment_ptr(batch, size), batch.contiguous())


# ------------------------------------------------------------ ReLU + dropout (fused) --
class _DropoutState:
    """Device step counter + per-step call-site numbering for the fused activation's generator."""

    def __init__(self):
        self.step = {}           # device -> int64[1]
        self.site = 0
        self.seed = None

    def counter(self, dev):
        t = self.step.get(dev)
        if t is None:
            t = self.step[dev] = torch.zeros(1, dtype=torch.int64, device=dev)
        return t

    def next_step(self, dev):
        """Called by the trainers once per training step (inside the captured graph as well)."""
        self.counter(dev).add_(1)
        self.site = 0

    def next_site(self):
        self.site += 1
        return self.site


dropout_state = _DropoutState()


class _ReluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        x = _f32c(x, "x")
        y = torch.empty_like(x)
        st = dropout_state
        if st.seed is None:
            st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        L = _lib.lib()
        _lib.check(L.gda_relu_dropout_fwd_f32(_lib.ptr(x), _lib.ptr(y), x.numel(), float(p),
                                              ctypes.c_uint64(st.seed), _lib.ptr(st.counter(x.device)),
                                              ctypes.c_uint32(st.next_site()), _lib.stream()),
                   "gda_relu_dropout_fwd_f32")
        ctx.save_for_backward(y)
        ctx.p = float(p)
        ctx.sink = _offer_sink(y, p, wanted=ctx.needs_input_grad[0])
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        if ctx.sink is not None and ctx.sink.mine(gy):       # the producer stored it masked already (GradSink)
            return gy, None
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        L = _lib.lib()
        _lib.check(L.gda_relu_dropout_bwd_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(gx), gy.numel(), ctx.p,
                                              _lib.stream()), "gda_relu_dropout_bwd_f32")
        return gx, None


class _ReluDropoutT(torch.autograd.Function):
    """The activation across the layout change: column-major input ``[d, n_pad]`` -> row-major ``[n, d]``
    output, and back in the backward pass (gda_relu_dropout_{fwd,bwd}_cm_f32)."""

    @staticmethod
    def forward(ctx, xT, n, p):
        d, n_pad = xT.shape
        y = torch.empty(n, d, dtype=torch.float32, device=xT.device)
        st = dropout_state
        if st.seed is None:
            st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        L = _lib.lib()
        _lib.check(L.gda_relu_dropout_fwd_cm_f32(_lib.ptr(xT), n_pad, _lib.ptr(y), n, d, float(p),
                                                ctypes.c_uint64(st.seed), _lib.ptr(st.counter(xT.device)),
                                                ctypes.c_uint32(st.next_site()), _lib.stream()),
                   "gda_relu_dropout_fwd_cm_f32")
        ctx.save_for_backward(y)
        ctx.p, ctx.n_pad = float(p), n_pad
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        n, d = y.shape
        gy = gy.contiguous()
        gxT = torch.empty(d, ctx.n_pad, dtype=torch.float32, device=gy.device)
        if ctx.n_pad != n:
            gxT[:, n:].zero_()                   # the padding rows of a column are never read as values,
        L = _lib.lib()                           # but they may be summed with another consumer's gradient
        _lib.check(L.gda_relu_dropout_bwd_cm_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(gxT), ctx.n_pad, n, d, ctx.p,
                                                _lib.stream()), "gda_relu_dropout_bwd_cm_f32")
        return gxT, None, None


class _ReluDropoutPair(torch.autograd.Function):
    """gda_relu_dropout_pair_{fwd,bwd}_f32: ``[drop_a(relu(x)) ; drop_b(relu(x))]`` ``[2n, d]``."""

    @staticmethod
    def forward(ctx, x, p):
        x = _f32c(x, "x")
        n, d = x.shape
        y = torch.empty(2 * n, d, dtype=torch.float32, device=x.device)
        st = dropout_state
        if st.seed is None:
            st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        L = _lib.lib()
        _lib.check(L.gda_relu_dropout_pair_fwd_f32(_lib.ptr(x), _lib.ptr(y), n, d, float(p), ctypes.c_uint64(st.seed),
                                                   _lib.ptr(st.counter(x.device)), ctypes.c_uint32(st.next_site()),
                                                   ctypes.c_uint32(st.next_site()), _lib.stream()),
                   "gda_relu_dropout_pair_fwd_f32")
        ctx.save_for_backward(y)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        n, d = y.size(0) // 2, y.size(1)
        gy = gy.contiguous()
        gx = torch.empty(n, d, dtype=torch.float32, device=gy.device)
        L = _lib.lib()
        cs = ws = None
        nbytes = 0
        if d <= 1024:                     # column sums of gx ride along: the bias gradient of the layer below
            cs = torch.empty(d, dtype=torch.float32, device=gy.device)
            nbytes = L.gda_relu_dropout_pair_workspace_bytes(d)
            ws = _lib.workspace(nbytes, gy.device, "pair_colsum")
        _lib.check(L.gda_relu_dropout_pair_bwd_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(gx), n, d, ctx.p, _lib.ptr(cs),
                                                   _lib.ptr(ws), nbytes, _lib.stream()), "gda_relu_dropout_pair_bwd_f32")
        global _colsum_hint
        _colsum_hint = (gx.data_ptr(), cs, gx) if cs is not None else None
        return gx, None


def relu_dropout_pair_ok(x):
    return torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.size(1) % 4 == 0


def relu_dropout_pair(x, p, training=True):
    """Two independent ``F.dropout(F.relu(x), p)`` draws of the same ``x [n, d]`` as one stacked ``[2n, d]``."""
    return _ReluDropoutPair.apply(x, float(p) if training else 0.0)


class _SplitHalves(torch.autograd.Function):
    """``x [2n, d] -> (x[:n], x[n:])``; backward stacks the two gradients in one launch (a missing one reads as
    zeros) instead of autograd's fill + copy + add per slice."""

    @staticmethod
    def forward(ctx, x):
        n = x.size(0) // 2
        ctx.shape = (n, x.size(1))
        ctx.set_materialize_grads(False)
        return x.narrow(0, 0, n), x.narrow(0, n, n)

    @staticmethod
    def backward(ctx, ga, gb):
        n, d = ctx.shape
        ref = ga if ga is not None else gb
        if ref is None:
            return None
        out = torch.empty(2 * n, d, dtype=torch.float32, device=ref.device)
        ga = None if ga is None else ga.contiguous()
        gb = None if gb is None else gb.contiguous()
        _lib.check(_lib.lib().gda_stack2_f32(_lib.ptr(ga), _lib.ptr(gb), _lib.ptr(out), n * d, _lib.stream()),
                   "gda_stack2_f32")
        return out


def split_halves(x):
    return _SplitHalves.apply(x)


class _SplitRows(torch.autograd.Function):
    """``x [n, d] -> (x[:k], x[k:])`` (views); backward writes the two gradients into ONE buffer (a missing one as
    zeros): two copies instead of autograd's fill + copy per slice and the add that joins them."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k, ctx.shape = int(k), tuple(x.shape)
        ctx.set_materialize_grads(False)
        return x.narrow(0, 0, ctx.k), x.narrow(0, ctx.k, x.size(0) - ctx.k)

    @staticmethod
    def backward(ctx, ga, gb):
        ref = ga if ga is not None else gb
        if ref is None:
            return None, None
        out = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        for part, g in ((out[:ctx.k], ga), (out[ctx.k:], gb)):
            if g is None:
                part.zero_()
            else:
                part.copy_(g)
        return out, None


def split_rows(x, k):
    """The first ``k`` rows and the rest of a stacked matrix (a block-diagonal pair of graphs: BaseGDA._stacked_pair)."""
    return _SplitRows.apply(x, k)


class _ReluDropoutSplit(torch.autograd.Function):
    """``split_halves(relu_dropout(x, p))`` for a stacked ``x [2n, d]``: the forward is the ordinary activation kernel,
    the backward masks the two halves' gradients while it stacks them (gda_relu_dropout_bwd2_f32) -- the unmasked
    ``[2n, d]`` stack that `_SplitHalves` + `_ReluDropout` write and read back is never formed."""

    @staticmethod
    def forward(ctx, x, p):
        x = _f32c(x, "x")
        y = torch.empty_like(x)
        st = dropout_state
        if st.seed is None:
            st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        _lib.check(_lib.lib().gda_relu_dropout_fwd_f32(_lib.ptr(x), _lib.ptr(y), x.numel(), float(p),
                                                       ctypes.c_uint64(st.seed), _lib.ptr(st.counter(x.device)),
                                                       ctypes.c_uint32(st.next_site()), _lib.stream()),
                   "gda_relu_dropout_fwd_f32")
        ctx.save_for_backward(y)
        ctx.p = float(p)
        ctx.set_materialize_grads(False)
        n = x.size(0) // 2
        a, b = y.narrow(0, 0, n), y.narrow(0, n, n)
        ctx.sinks, ctx.G = (None, None), None
        if _sinks_on and ctx.needs_input_grad[0]:
            # both halves' masked gradients land in ONE [2n, d] buffer: when both producers wrote, it IS the result
            ctx.G = torch.empty_like(y)
            ctx.sinks = (_offer_sink(a, p, ctx.G.narrow(0, 0, n)), _offer_sink(b, p, ctx.G.narrow(0, n, n)))
        return a, b

    @staticmethod
    def backward(ctx, ga, gb):
        (y,) = ctx.saved_tensors
        if ga is None and gb is None:
            return None, None
        L = _lib.lib()
        n = y.size(0) // 2
        done = [k is not None and k.mine(g) for k, g in zip(ctx.sinks, (ga, gb))]
        if any(done):
            # a half that arrived masked sits in G already; the other one (an ordinary gradient, or none) is masked into its
            # half of G by the ordinary kernel
            for k, (g, ok) in enumerate(zip((ga, gb), done)):
                if ok:
                    continue
                dst, yh = ctx.G.narrow(0, k * n, n), y.narrow(0, k * n, n)
                if g is None:
                    dst.zero_()
                else:
                    g = g.contiguous()
                    _lib.check(L.gda_relu_dropout_bwd_f32(_lib.ptr(g), _lib.ptr(yh), _lib.ptr(dst), g.numel(), ctx.p,
                                                          _lib.stream()), "gda_relu_dropout_bwd_f32")
            return ctx.G, None
        ga = None if ga is None else ga.contiguous()
        gb = None if gb is None else gb.contiguous()
        gx = torch.empty_like(y)
        _lib.check(L.gda_relu_dropout_bwd2_f32(_lib.ptr(ga), _lib.ptr(gb), _lib.ptr(y), _lib.ptr(gx),
                                               y.numel() // 2, ctx.p, _lib.stream()), "gda_relu_dropout_bwd2_f32")
        return gx, None


# ------------------------------- projections of a sampled batch: gather in the operand fetch, activation in the epilogue --
TALL_FUSED = _os.environ.get("PYGDA_AMD_TALL_FUSED", "1") == "1"


class GatheredRows:
    """``base[idx]`` NOT materialised: the feature rows of a sampled batch as (resident feature matrix, node ids).  The
    batch's first projection reads them through the ids in its operand fetch (gda_gemm_tall_fwd_ex_f32) and so does its
    weight gradient (gda_gemm_tall_wgrad_gather_f32): the gather pass that wrote and re-read ``[n, F]`` per domain and step
    is gone.  Quacks like the tensor where the trainers only ask for its shape / device; ``dense()`` materialises."""
    __slots__ = ("base", "idx", "_dense")
    requires_grad = False
    _version = 0

    def __init__(self, base, idx):
        self.base, self.idx, self._dense = base, idx, None

    shape = property(lambda self: torch.Size((self.idx.numel(), self.base.size(1))))
    device = property(lambda self: self.base.device)
    dtype = property(lambda self: self.base.dtype)
    is_cuda = property(lambda self: self.base.is_cuda)

    def size(self, k=None):
        return self.shape if k is None else self.shape[k]

    def dim(self):
        return 2

    def data_ptr(self):
        return 0

    def dense(self):
        if self._dense is None:
            self._dense = gather_rows(self.base, self.idx)
        return self._dense


def tall_fused_ok(x, weight, need_rows=None):
    """The envelope of gda_gemm_tall_fwd_ex_f32 for ``x W^T``: a sampled batch's row count, extents 128 / 256."""
    rows = x.size(0)
    return (TALL_FUSED and x.is_cuda and x.dtype == torch.float32 and rows >= TALL_ROWS
            and weight.size(0) in (128, 256) and weight.size(1) in (128, 256) and x.size(1) == weight.size(1)
            and _os.environ.get("PYGDA_AMD_GEMM_SPLIT_F16", "1") != "0")


class _TallLinearAct(torch.autograd.Function):
    """``act(x W^T + b)`` of a sampled batch in ONE launch: the tall split-fp16 product with (optionally) the feature
    gather in its operand fetch and ``dropout(relu(.))`` in its epilogue.  ``mode``: 0 no activation; 1 activation; 2 two
    independent draws stacked ``[2M, N]`` (A2GNN's two source passes over one layer-0 output); 3 activation, result handed
    out as its two halves (the stacked rows' last layer), both halves offering GradSinks into one ``[M, N]`` buffer.
    Backward: the activation's mask from the saved OUTPUT, the weight gradient through the same node ids, the bias
    gradient as the weight-gradient kernel's column sums (mode 2: of the pair kernel's)."""

    @staticmethod
    def forward(ctx, base, idx, weight, bias, p, mode):
        base, weight = _f32c(base, "x"), _f32c(weight, "weight")
        M = base.size(0) if idx is None else idx.numel()
        K, N = base.size(1), weight.size(0)
        y = torch.empty((2 * M if mode == 2 else M), N, dtype=torch.float32, device=base.device)
        st = dropout_state
        if st.seed is None:
            st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        site0 = st.next_site() if mode else 0
        site1 = st.next_site() if mode == 2 else 0
        b = None if bias is None else _f32c(bias, "bias")
        L = _lib.lib()
        with profiler.region(f"dense_projection[{K}x{N}]", 1, 4 * (M * K + N * K + y.numel()), 2 * M * N * K):
            _lib.check(L.gda_gemm_tall_fwd_ex_f32(M, N, K, _lib.ptr(base), K, _lib.ptr(idx), _lib.ptr(weight), K, _lib.ptr(y), N,
                                                  _lib.ptr(b), 2 if mode == 2 else (1 if mode else 0), float(p),
                                                  ctypes.c_uint64(st.seed), _lib.ptr(st.counter(base.device)),
                                                  ctypes.c_uint32(site0), ctypes.c_uint32(site1), _lib.stream()),
                       "gda_gemm_tall_fwd_ex_f32")
        ctx.mode, ctx.p, ctx.has_bias, ctx.M = mode, float(p), bias is not None, M
        ctx.sink_in = sink_of(base) if idx is None else None
        ctx.save_for_backward(base, idx, weight, y if mode else None)
        want = ctx.needs_input_grad[0] or ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        ctx.sink = ctx.sinks = ctx.G = None
        if mode == 1:
            ctx.sink = _offer_sink(y, p, wanted=want)
        if mode == 3:
            h = M // 2
            a, c = y.narrow(0, 0, h), y.narrow(0, h, h)
            if _sinks_on and want:
                ctx.G = torch.empty_like(y)
                ctx.sinks = (_offer_sink(a, p, ctx.G.narrow(0, 0, h)), _offer_sink(c, p, ctx.G.narrow(0, h, h)))
            ctx.set_materialize_grads(False)
            return a, c
        return y

    @staticmethod
    def backward(ctx, g, g2=None):
        base, idx, weight, y = ctx.saved_tensors
        L = _lib.lib()
        M, mode, dev = ctx.M, ctx.mode, weight.device
        N, K = weight.shape
        gb = None
        if mode == 0:
            gpre = g.contiguous()
        elif mode == 1:
            if ctx.sink is not None and ctx.sink.mine(g):
                gpre = g
            else:
                g = g.contiguous()
                gpre = torch.empty_like(g)
                _lib.check(L.gda_relu_dropout_bwd_f32(_lib.ptr(g), _lib.ptr(y), _lib.ptr(gpre), g.numel(), ctx.p, _lib.stream()),
                           "gda_relu_dropout_bwd_f32")
        elif mode == 2:
            # (the bias gradient comes from the weight-gradient kernel's column sums below, as in tall_linear_bias: the same
            # summation order as the unfused composition)
            g = g.contiguous()
            gpre = torch.empty(M, N, dtype=torch.float32, device=dev)
            _lib.check(L.gda_relu_dropout_pair_bwd_f32(_lib.ptr(g), _lib.ptr(y), _lib.ptr(gpre), M, N, ctx.p, None,
                                                       None, 0, _lib.stream()), "gda_relu_dropout_pair_bwd_f32")
        else:
            ga, gc = g, g2
            if ga is None and gc is None:
                return None, None, None, None, None, None
            h = M // 2
            done = [k is not None and k.mine(t) for k, t in zip(ctx.sinks or (None, None), (ga, gc))]
            if any(done):
                for k, (t, ok) in enumerate(zip((ga, gc), done)):
                    if ok:
                        continue
                    dst, yh = ctx.G.narrow(0, k * h, h), y.narrow(0, k * h, h)
                    if t is None:
                        dst.zero_()
                    else:
                        t = t.contiguous()
                        _lib.check(L.gda_relu_dropout_bwd_f32(_lib.ptr(t), _lib.ptr(yh), _lib.ptr(dst), t.numel(), ctx.p,
                                                              _lib.stream()), "gda_relu_dropout_bwd_f32")
                gpre = ctx.G
            else:
                ga = None if ga is None else ga.contiguous()
                gc = None if gc is None else gc.contiguous()
                gpre = torch.empty_like(y)
                _lib.check(L.gda_relu_dropout_bwd2_f32(_lib.ptr(ga), _lib.ptr(gc), _lib.ptr(y), _lib.ptr(gpre), y.numel() // 2,
                                                       ctx.p, _lib.stream()), "gda_relu_dropout_bwd2_f32")
        gw = None
        if ctx.needs_input_grad[2]:
            want_cs = ctx.has_bias and gb is None and ctx.needs_input_grad[3]
            if want_cs:
                gb = torch.empty(N, dtype=torch.float32, device=dev)
            if idx is None:
                gw = gemm(GEMM_TN, gpre, base, colsum=gb if want_cs else None)
            else:
                gw = torch.empty(N, K, dtype=torch.float32, device=dev)
                need = L.gda_gemm_tall_workspace_bytes(GEMM_TN, 128, K, M)
                ok = N == 128 and need > 0 and M >= TALL_WGRAD_ROWS
                if ok:
                    ws = _lib.workspace(need, dev, "gemm")
                    with profiler.region(f"dense_projection_wgrad[{N}x{K}]", 2, 4 * (M * N + M * K + N * K), 2 * M * N * K):
                        _lib.check(L.gda_gemm_tall_wgrad_gather_f32(K, M, _lib.ptr(gpre), N, _lib.ptr(base), K, _lib.ptr(idx),
                                                                    _lib.ptr(gw), K, _lib.ptr(gb if want_cs else None),
                                                                    _lib.ptr(ws), ws.numel(), _lib.stream()),
                                   "gda_gemm_tall_wgrad_gather_f32")
                else:
                    gw = gemm(GEMM_TN, gpre, gather_rows(base, idx), colsum=gb if want_cs else None)
        elif ctx.has_bias and gb is None and ctx.needs_input_grad[3]:
            gb = colsum(gpre)
        gx = None
        if idx is None and ctx.needs_input_grad[0]:
            gx = masked_dgrad(gpre, weight, ctx.sink_in) if ctx.sink_in is not None else gemm(GEMM_NN, gpre, weight)
        return gx, None, gw, (gb if ctx.has_bias else None), None, None


def tall_linear_act(x, weight, bias, p, training, mode):
    """See :class:`_TallLinearAct`; ``x`` a tensor or :class:`GatheredRows`; the caller checked :func:`tall_fused_ok`."""
    base, idx = (x.base, x.idx) if isinstance(x, GatheredRows) else (x, None)
    return _TallLinearAct.apply(base, idx, weight, bias, float(p) if training else 0.0, int(mode))


def relu_dropout_split(x, p, training=True):
    """``split_halves(relu_dropout(x, p, training))`` with a fused backward; ``x [2n, d]`` on the device, d % 4 == 0,
    training with p > 0 -- otherwise the composition."""
    if (training and p > 0.0 and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
            and x.size(0) % 2 == 0 and (x.size(0) // 2 * x.size(1)) % 4 == 0):
        return _ReluDropoutSplit.apply(x, float(p))
    return split_halves(relu_dropout(x, p, training))


def relu_dropout_copies(x, copies, p, training=True):
    """``F.dropout(F.relu(x.repeat(copies, 1)), p, training)`` for ``x [n, d]`` without materialising the copies
    (include/gda_hip.h: gda_relu_dropout_tiled_fwd_f32); forward only."""
    if torch.is_grad_enabled() and x.requires_grad:
        raise _lib.GdaError("relu_dropout_copies is forward-only: call it under torch.no_grad()")
    if (not training or p <= 0.0 or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.numel() % 4 != 0
            or not x.is_contiguous() or x.data_ptr() % 16 != 0):
        return relu_dropout(x.repeat(copies, 1), p, training)
    y = torch.empty(copies * x.size(0), x.size(1), dtype=torch.float32, device=x.device)
    st = dropout_state
    if st.seed is None:
        st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
    _lib.check(_lib.lib().gda_relu_dropout_tiled_fwd_f32(
        _lib.ptr(x), x.numel(), int(copies), _lib.ptr(y), float(p), ctypes.c_uint64(st.seed), _lib.ptr(st.counter(x.device)),
        ctypes.c_uint32(st.next_site()), _lib.stream()), "gda_relu_dropout_tiled_fwd_f32")
    return y


def relu_dropout(x, p, training=True):
    """``F.dropout(F.relu(x), p, training)`` in one kernel each way (no mask tensor).  A :class:`ColMajor`
    input (the K-step kernel's layout) is consumed as it is and comes out as the ordinary row-major tensor."""
    if isinstance(x, ColMajor):
        return _ReluDropoutT.apply(x.t, x.n, float(p) if training else 0.0)
    if not training or p <= 0.0 or not x.is_cuda or x.dtype != torch.float32:
        return torch.relu(x)
    return _ReluDropout.apply(x, p)


# ------------------------------------------ mixup layer epilogue (StruRW mode='mixup') --
class _MixupCombine(torch.autograd.Function):
    """gda_mixup_combine_{fwd,bwd}_f32: ``XX' = [drop(relu(P + C + b)) ; drop(lam relu(P + Cm + b) +
    (1-lam) relu(Pb + Cm + b))]`` with ``Pb = P[perm]`` unless given (pygda/nn/mixup_base.py:146-196)."""

    @staticmethod
    def forward(ctx, P, Pb, CC, bias, perm, inv, lam, p, first):
        P, CC, bias = _f32c(P, "P"), _f32c(CC, "CC"), _f32c(bias, "bias")
        if Pb is not None:
            Pb = _f32c(Pb, "Pb")
        n, h = P.shape
        if CC.shape != ((n, h) if first else (2 * n, h)):
            raise ValueError(f"centre projections must be {'[n, h]' if first else '[2n, h]'}, got {tuple(CC.shape)}")
        XX = torch.empty(2 * n, h, dtype=torch.float32, device=P.device)
        mask = torch.empty(n, h, dtype=torch.uint8, device=P.device)
        st = dropout_state
        if st.seed is None:
            st.seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        L = _lib.lib()
        _lib.check(L.gda_mixup_combine_fwd_f32(
            _lib.ptr(P), _lib.ptr(Pb), _lib.ptr(CC), int(first), _lib.ptr(bias), _lib.ptr(perm), n, h, float(lam),
            float(p), ctypes.c_uint64(st.seed), _lib.ptr(st.counter(P.device)), ctypes.c_uint32(st.next_site()),
            ctypes.c_uint32(st.next_site()), _lib.ptr(XX), _lib.ptr(mask), _lib.stream()), "gda_mixup_combine_fwd_f32")
        ctx.save_for_backward(XX, mask, inv)
        ctx.cfg = (float(lam), float(p), bool(first), Pb is not None)
        return XX

    @staticmethod
    def backward(ctx, gXX):
        XX, mask, inv = ctx.saved_tensors
        lam, p, first, sep = ctx.cfg
        n, h = mask.shape
        gXX = gXX.contiguous()
        f32 = dict(dtype=torch.float32, device=gXX.device)
        gP = torch.empty(n, h, **f32)
        gPb = torch.empty(n, h, **f32) if sep else None
        gCC = torch.empty(n if first else 2 * n, h, **f32)
        gb = torch.empty(h, **f32)
        L = _lib.lib()
        nbytes = L.gda_mixup_combine_workspace_bytes(n, h)
        ws = _lib.workspace(nbytes, gXX.device, "mixup")
        _lib.check(L.gda_mixup_combine_bwd_f32(
            _lib.ptr(gXX), _lib.ptr(XX), _lib.ptr(mask), _lib.ptr(inv), int(first), n, h, lam, p, _lib.ptr(gP),
            _lib.ptr(gPb), _lib.ptr(gCC), _lib.ptr(gb), _lib.ptr(ws), nbytes, _lib.stream()), "gda_mixup_combine_bwd_f32")
        return gP, gPb, gCC, gb, None, None, None, None, None


def mixup_combine_ok(P, h):
    return P.is_cuda and P.dtype == torch.float32 and h % 4 == 0 and h <= 1024


def mixup_combine(P, Pb, CC, bias, perm, inv, lam, p, training, first):
    """One mixup layer's epilogue -> the stacked pair ``[x' ; x_mix']`` ``[2n, h]``.  ``perm`` / ``inv`` are
    int64 device vectors (new position -> old node and its inverse)."""
    return _MixupCombine.apply(P, Pb, CC, bias, perm, inv, float(lam), float(p) if training else 0.0, bool(first))


# ------------------------------------------------- Laplacian smoothness (TDSS) --
class _Laplacian(torch.autograd.Function):
    """``1/2 sum_e ||f[row] dinv[row] - f[col] dinv[col]||^2`` over the edges of ``graph`` (built
    with ``add_self_loops=False, normalize=False``: by-source CSR = rows of ``row`` nodes)."""

    @staticmethod
    def forward(ctx, feats, graph):
        f = _f32c(feats, "features")
        n, d = f.shape
        if n != graph.num_nodes:
            raise ValueError(f"features must have num_nodes={graph.num_nodes} rows, got {n}")
        dev = f.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dinv = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = _lib.workspace(L.gda_laplacian_workspace_bytes(n), dev, "laplacian")
        nnz = graph.nnz if profiler.enabled else 0
        with profiler.region(f"laplacian_fwd[d={d}]", 1, nnz * 4 + n * (d * 4 + 12), 3 * nnz * d):
            _lib.check(L.gda_laplacian_fwd_f32(_lib.ptr(graph.t_rowptr), _lib.ptr(graph.t_colidx), n, d,
                                               _lib.ptr(f), d, _lib.ptr(loss), _lib.ptr(dinv), _lib.ptr(ws),
                                               ws.numel(), _lib.stream()), "gda_laplacian_fwd_f32")
        ctx.graph = graph
        ctx.save_for_backward(f, dinv)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        f, dinv = ctx.saved_tensors
        g = ctx.graph
        n, d = f.shape
        gf = torch.empty_like(f)
        gl = gl.reshape(1).to(torch.float32).contiguous()
        L = _lib.lib()
        nnz = g.nnz if profiler.enabled else 0
        with profiler.region(f"laplacian_bwd[d={d}]", 1, nnz * 8 + 2 * n * d * 4, 4 * nnz * d):
            _lib.check(L.gda_laplacian_bwd_f32(_lib.ptr(g.t_rowptr), _lib.ptr(g.t_colidx), _lib.ptr(g.rowptr),
                                               _lib.ptr(g.colidx), n, d, _lib.ptr(f), d, _lib.ptr(dinv),
                                               _lib.ptr(gl), _lib.ptr(gf), d, _lib.stream()),
                       "gda_laplacian_bwd_f32")
        return gf, None


def laplacian_loss(features, edge_index):
    """TDSS.compute_laplacian_loss (pygda/models/tdss.py:390-454).  ``edge_index`` is the
    ``[2, E]`` smoothing edge list (ingested once, cached by identity) or a ready
    :class:`~pygda_amd.graph.CSRGraph` built with ``add_self_loops=False, normalize=False``."""
    from .graph import as_graph
    graph = as_graph(edge_index, features.size(0), None, False, False, False, "col")
    return _Laplacian.apply(features, graph)


# ------------------------------------------- polynomial graph filters (DGSDA BernProp) --
def spmm_axpby(graph: CSRGraph, x, alpha, beta, z=None, gamma=1.0, gamma_dev=None, transposed=False, out=None):
    """``alpha * x + beta * (A x) + gamma * z`` in one aggregation launch (no autograd).
    ``gamma_dev``: one-element device tensor multiplied into ``gamma`` (a learnable coefficient);
    ``out``: contiguous ``[n, d]`` fp32 destination (e.g. a slice of a chain buffer)."""
    x = _f32c(x, "x")
    if x.dim() != 2 or x.size(0) != graph.num_nodes:
        raise ValueError(f"x must be [num_nodes={graph.num_nodes}, d], got {tuple(x.shape)}")
    rp, ci, va = (graph.t_rowptr, graph.t_colidx, graph.t_val) if transposed else \
                 (graph.rowptr, graph.colidx, graph.val)
    n, d = x.shape
    if z is not None:
        z = _f32c(z, "z")
        if z.shape != x.shape:
            raise ValueError("z must have the shape of x")
    if out is None:
        y = torch.empty_like(x)
    else:
        y = out
        if y.shape != x.shape or y.dtype != torch.float32 or not y.is_contiguous() or y.device != x.device:
            raise ValueError("out must be a contiguous fp32 tensor of the shape of x on its device")
    L = _lib.lib()
    if aggregation_log is not None:
        aggregation_log.append((_logged(graph), 1))
    if profiler.enabled:
        global aggregated_edges
        aggregated_edges += graph.nnz
        ctx = profiler.region(f"spmm_csr_axpby_f32[d={d}]", 1,
                              graph.nnz * 8 + (n + 1) * 4 + (2 + (z is not None)) * n * d * 4, 2 * graph.nnz * d)
    else:
        ctx = profiler.region("", 0)
    sp = graph.split(transposed).struct(d)
    with ctx:
        _lib.check(L.gda_spmm_csr_axpby_f32(_lib.ptr(rp), _lib.ptr(ci), _lib.ptr(va), n, d, _lib.ptr(x), d,
                                            _lib.ptr(y), d, float(alpha), float(beta), _lib.ptr(z), d,
                                            float(gamma), _lib.ptr(gamma_dev),
                                            ctypes.byref(sp) if sp is not None else None, _lib.stream()),
                   "gda_spmm_csr_axpby_f32")
    return y


class _BernFilter(torch.autograd.Function):
    """``out = sum_k w_k (I + A)^(K-k) (I - A)^k x`` with ``w = relu(temp) * binom(K, k) / 2^K``: the
    Bernstein-basis filter of BernProp.forward (pygda/nn/dgsda_base.py:101-151), where
    L = I - A and 2I - L = I + A for the self-loop-free symmetric normalisation A.

    The reference spends K + K(K+1)/2 propagations forward (and as many again in autograd).  L and
    2I - L commute, so the same polynomial is evaluated as one chain v_k = L^k x and one Horner
    sweep s_j = (I + A) s_{j-1} + w_j v_j: 2K launches forward, 2K backward (adjoint chain
    q_j = (I + A^T)^j g, Horner in (I - A^T)), the accumulation riding in the SpMM epilogue and the
    coefficients read from device memory.  Every term is a product of positive semi-definite
    factors with non-negative weights, so the reordering costs rounding only (tested at 1e-5)."""

    @staticmethod
    def forward(ctx, x, temp, graph, coefs):
        K = temp.numel() - 1
        w = (torch.relu(temp.detach()) * coefs).contiguous()
        x = _f32c(x.detach(), "x")
        v = torch.empty((K + 1,) + tuple(x.shape), dtype=torch.float32, device=x.device)   # the chain L^k x
        v[0].copy_(x)
        for k in range(1, K + 1):
            spmm_axpby(graph, v[k - 1], 1.0, -1.0, out=v[k])
        s = v[0] * w[0]
        for j in range(1, K + 1):
            s = spmm_axpby(graph, s, 1.0, 1.0, z=v[j], gamma=1.0, gamma_dev=w[j:j + 1])
        ctx.graph, ctx.K = graph, K
        ctx.save_for_backward(temp, coefs, w, v)
        return s

    @staticmethod
    def backward(ctx, g):
        temp, coefs, w, v = ctx.saved_tensors
        graph, K = ctx.graph, ctx.K
        q = torch.empty_like(v)                     # q[K - j] = (I + A^T)^j g: slot k pairs with v[k]
        q[K].copy_(g)
        for j in range(1, K + 1):
            spmm_axpby(graph, q[K - j + 1], 1.0, 1.0, transposed=True, out=q[K - j])
        g_temp = None
        if ctx.needs_input_grad[1]:                 # <(I+A^T)^(K-k) g, L^k x> for all k at once
            dots = (q.view(K + 1, -1) * v.view(K + 1, -1)).sum(dim=1)
            g_temp = dots * coefs * (temp > 0).to(dots.dtype)
        g_x = None
        if ctx.needs_input_grad[0]:
            t = q[K] * w[K]
            for j in range(1, K + 1):
                t = spmm_axpby(graph, t, 1.0, -1.0, z=q[K - j], gamma=1.0, gamma_dev=w[K - j:K - j + 1],
                               transposed=True)
            g_x = t
        return g_x, g_temp, None, None


def bern_filter(x, temp, graph: CSRGraph, coefs):
    return _BernFilter.apply(x, temp, graph, coefs)


# ------------------------------------------------- view attention (UDAGCN's dual-view encoder) --
class _AttentionFuse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, b, *views):
        L = _lib.lib()
        xs = [_f32c(v, "view") for v in views]
        n, h = xs[0].shape
        K = len(xs)
        w1 = _f32c(w, "dense_weight.weight").reshape(-1)
        b1 = _f32c(b, "dense_weight.bias").reshape(-1)
        out = torch.empty(n, h, dtype=torch.float32, device=xs[0].device)
        att = torch.empty(n, K, dtype=torch.float32, device=xs[0].device)
        ptrs = (ctypes.c_void_p * K)(*[x.data_ptr() for x in xs])
        lds = (ctypes.c_int64 * K)(*[h] * K)
        _lib.check(L.gda_attention_fuse_fwd_f32(K, ptrs, lds, n, h, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(out), h,
                                                _lib.ptr(att), _lib.stream()), "gda_attention_fuse_fwd_f32")
        ctx.save_for_backward(w1, att, *xs)
        ctx.w_shape, ctx.b_shape = w.shape, b.shape
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        w1, att, *xs = ctx.saved_tensors
        n, h = xs[0].shape
        K = len(xs)
        g = _f32c(g, "grad")
        gxs = [torch.empty_like(x) if ctx.needs_input_grad[2 + k] else None for k, x in enumerate(xs)]
        gw = torch.empty(h, dtype=torch.float32, device=g.device)
        gb = torch.empty(1, dtype=torch.float32, device=g.device)
        ws = _lib.workspace(L.gda_attention_workspace_bytes(n, h), g.device, "attention")
        ptrs = (ctypes.c_void_p * K)(*[x.data_ptr() for x in xs])
        gptrs = (ctypes.c_void_p * K)(*[None if t is None else t.data_ptr() for t in gxs])
        lds = (ctypes.c_int64 * K)(*[h] * K)
        _lib.check(L.gda_attention_fuse_bwd_f32(K, ptrs, lds, n, h, _lib.ptr(w1), _lib.ptr(att), _lib.ptr(g), h, gptrs,
                                                _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), ws.numel() if ws is not None else 0,
                                                _lib.stream()), "gda_attention_fuse_bwd_f32")
        return (gw.view(ctx.w_shape), gb.view(ctx.b_shape), *gxs)


def attention_fuse_ok(views, w):
    """Shapes the fused view-attention kernels cover (csrc/gda_attention.hip)."""
    if not (2 <= len(views) <= 4):
        return False
    v0 = views[0]
    return (all(torch.is_tensor(v) and v.is_cuda and v.dtype == torch.float32 and v.dim() == 2 and v.shape == v0.shape
                for v in views) and v0.size(1) % 4 == 0 and 0 < v0.size(1) <= 512 and w.numel() == v0.size(1))


def attention_fuse(views, w, b):
    """``sum(stack(views, 1) * softmax(Linear(h, 1)(stack(views, 1)), 1), 1)`` (pygda/nn/attention.py:51-54), fused."""
    return _AttentionFuse.apply(w, b, *views)

