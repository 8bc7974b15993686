"""Device-resident normalised adjacency (CSR by destination + CSR by source).

The reference recomputes ``gcn_norm`` on every conv call (``cached=False``,
pygda/nn/prop_gcn_conv.py:182-192 -- 10x per A2GNN step) and lets PyG scatter over a COO
edge list.  Here the edge list is ingested once per graph / mini-batch by
``gda_build_csr_norm`` into the layout the SpMM kernel wants, and looked up again through
a small identity-keyed cache, so the ``forward(x, edge_index, ...)`` operator surface
stays unchanged.
"""
import weakref
from collections import OrderedDict

import torch

from . import _lib


import os

# rows with more entries than this are cut into chunks (power-law hubs)
SPLIT_THRESHOLD = int(os.environ.get("PYGDA_AMD_SPLIT_THRESHOLD", "128"))


class RowSplit:
    """Long-row chunk layout of one CSR (see ``gda_row_split`` in include/gda_hip.h)."""

    def __init__(self, rowptr, num_rows, nnz_cap, threshold=SPLIT_THRESHOLD, deferred=False):
        """``deferred``: never read the counts back (no host sync): launches are sized by the capacity
        bounds and the kernels pick up the live counts from device memory -- for graphs that live one
        step, where a sync per graph would drain the queue every mini-batch."""
        dev = rowptr.device
        self.threshold, self.counts, self._scratch = threshold, None, {}
        if not deferred and num_rows > 0:
            # a static matrix whose longest row is within 2 x the threshold gains nothing from chunking (a 131-entry row
            # beside 128-entry ones is no load imbalance) and pays for it with the chunk reduce launch on every call --
            # the bag-of-words transpose of the cfg-A stand-ins (rows ~ Binomial(9360, 0.01): longest ~131) was split
            # for a handful of such rows; whole rows also keep the sequential, CPU-order sums
            longest = int((rowptr[1:num_rows + 1] - rowptr[:num_rows]).max())
            if longest <= 2 * threshold:
                self.n_long = self.n_chunks = 0
                return
        cap_long = nnz_cap // threshold + 1
        cap_chunks = 2 * (nnz_cap // threshold) + 2
        i32 = dict(dtype=torch.int32, device=dev)
        self.long_rows = torch.empty(cap_long, **i32)
        self.long_chunk_ptr = torch.empty(cap_long + 1, **i32)
        self.chunk_long = torch.empty(cap_chunks, **i32)
        counts = torch.zeros(2, **i32)
        L = _lib.lib()
        ws = _lib.workspace(L.gda_row_split_workspace_bytes(num_rows), dev, "split")
        _lib.check(L.gda_row_split_build(_lib.ptr(rowptr), num_rows, threshold, _lib.ptr(self.long_rows),
                                         _lib.ptr(self.long_chunk_ptr), _lib.ptr(self.chunk_long),
                                         _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream()),
                   "gda_row_split_build")
        self.counts = counts if deferred else None
        if deferred:
            self.n_long, self.n_chunks = cap_long, cap_chunks
        else:
            self.n_long, self.n_chunks = (int(v) for v in counts.tolist())   # one sync per graph

    def struct(self, d):
        """The C struct for a width-``d`` call (scratch grown on demand), or None without hubs."""
        if self.n_long == 0:
            return None
        need = self.n_chunks * d
        key = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < need:
            buf = self._scratch[key] = torch.empty(need, dtype=torch.float32, device=self.long_rows.device)
        return _lib.RowSplitStruct(self.threshold, self.n_long, self.n_chunks, self.long_rows.data_ptr(),
                                   self.long_chunk_ptr.data_ptr(), self.chunk_long.data_ptr(), buf.data_ptr(),
                                   self.counts.data_ptr() if self.counts is not None else None)


class CSRGraph:
    """``rowptr/colidx/val``: rows = destination nodes (forward aggregation);
    ``t_rowptr/t_colidx/t_val``: rows = source nodes (the transpose, backward)."""

    __slots__ = ("num_nodes", "nnz_cap", "rowptr", "colidx", "val", "t_rowptr", "t_colidx",
                 "t_val", "_nnz", "device", "_split", "_t_split", "t_to_fwd", "static", "_squared", "transient",
                 "_kplan", "_t_kplan", "n_interior", "iplan", "iplan_T", "_slot", "_gen", "tag")

    def __init__(self, num_nodes, nnz_cap, rowptr, colidx, val, t_rowptr, t_colidx, t_val):
        self.num_nodes, self.nnz_cap = num_nodes, nnz_cap
        self.rowptr, self.colidx, self.val = rowptr, colidx, val
        self.t_rowptr, self.t_colidx, self.t_val = t_rowptr, t_colidx, t_val
        self._nnz = None
        self.device = rowptr.device
        self._split = self._t_split = None
        self.t_to_fwd = None      # by-source entry -> by-destination position (built on request)
        self.static = False       # graph of a full-batch loader: lives for the whole fit()
        self.transient = False    # graph of one sampled mini-batch: nothing about it is worth a host sync
        self._squared = None      # A*A (and its transpose) of a static graph, or False if too dense
        self.tag = None           # bookkeeping label (the captured sampled step tags its two static graphs)
        self.iplan_T = None       # sampled batch: off-diagonal entries of the interior block (the sampler's count)
        self._slot, self._gen = None, 0        # sampled batch out of a recycling loader ring: its block and the block's
                                               # generation when this batch was written (as_graph refuses a stale one)
        self._kplan = self._t_kplan = None     # register programs of the LDS-resident K-step kernel, or False
        self.iplan = None         # sampled batch: (forward, transposed) device plans of the one-launch interior K-step
                                  # (csrc/gda_interior.inc; built by the device sampler), None where not eligible
        self.n_interior = None    # sampled batch: rows from here on hold their unit self loop only (the last hop's
                                  # discoveries are never expanded) and every row is short -- ops.spmm_kstep then
                                  # recomputes the interior rows only (gda_spmm_csr_interior_kstep_f32)

    def squared(self):
        """``A*A`` as a :class:`CSRGraph` (forward CSR from the forward CSR, transposed CSR from the
        transposed one), built once on the host (csrc/gda_smooth.cpp) -- or None when the product
        would hold more than ``SQUARE_MAX_FILL`` x the entries of A (power-law graphs)."""
        if self._squared is None:
            self._squared = _square(self) or False
        return self._squared or None

    def kstep_plan(self, transposed=False):
        """``(plan, slots)`` for the one-launch LDS-resident K-step kernel (csrc/gda_kstep.hip), compiled on
        the host from this CSR on first use (one device-to-host copy of the graph: static graphs only), or
        None when the graph is not eligible (more than 16,320 nodes, a row beyond 48 entries, ...)."""
        hit = self._t_kplan if transposed else self._kplan
        if hit is None:
            hit = _kstep_plan(self, transposed) or False
            if transposed:
                self._t_kplan = hit
            else:
                self._kplan = hit
        return hit or None

    def split(self, transposed=False):
        """Long-row layout of the forward (or transposed) CSR, built on first use."""
        if transposed:
            if self._t_split is None:
                self._t_split = RowSplit(self.t_rowptr, self.num_nodes, self.nnz_cap, deferred=self.transient)
            return self._t_split
        if self._split is None:
            self._split = RowSplit(self.rowptr, self.num_nodes, self.nnz_cap, deferred=self.transient)
        return self._split

    @property
    def nnz(self):
        """Number of stored entries incl. self loops (one host sync, then cached)."""
        if self._nnz is None:
            self._nnz = int(self.rowptr[self.num_nodes].item())
        return self._nnz

    def transposed(self):
        g = CSRGraph(self.num_nodes, self.nnz_cap, self.t_rowptr, self.t_colidx, self.t_val,
                     self.rowptr, self.colidx, self.val)
        g._split, g._t_split = self._t_split, self._split
        return g                  # (n_interior is a property of the by-destination orientation: not carried over)

    def to_coo(self):
        """(edge_index [2, nnz], weight [nnz]) in CSR order -- what gcn_norm returns."""
        nnz = self.nnz
        src = torch.empty(max(nnz, 1), dtype=torch.int64, device=self.device)
        dst = torch.empty(max(nnz, 1), dtype=torch.int64, device=self.device)
        L = _lib.lib()
        _lib.check(L.gda_csr_to_coo(_lib.ptr(self.rowptr), _lib.ptr(self.colidx), self.num_nodes, nnz,
                                    _lib.ptr(src), _lib.ptr(dst), _lib.stream()), "gda_csr_to_coo")
        return torch.stack([src[:nnz], dst[:nnz]]), self.val[:nnz]


def block_diag(a, b):
    """Two ingested graphs as ONE block-diagonal :class:`CSRGraph` (rows / columns of ``b`` behind those of ``a``): each
    row keeps its entries, their order and their values -- whatever normalisation either graph was ingested with --, so an
    aggregation over the pair is the two aggregations row for row, bit for bit (BaseGDA._stacked_pair: trainers that run
    one network over both domains; cached operators such as UDAGCN's PPMI graphs are combined, not rebuilt)."""
    na, nb, ea, eb = a.num_nodes, b.num_nodes, a.nnz, b.nnz

    def join(rp_a, ci_a, v_a, rp_b, ci_b, v_b):
        return (torch.cat([rp_a[:na + 1], rp_b[1:nb + 1] + ea]),
                torch.cat([ci_a[:ea], ci_b[:eb] + na]), torch.cat([v_a[:ea], v_b[:eb]]))

    g = CSRGraph(na + nb, ea + eb, *join(a.rowptr, a.colidx, a.val, b.rowptr, b.colidx, b.val),
                 *join(a.t_rowptr, a.t_colidx, a.t_val, b.t_rowptr, b.t_colidx, b.t_val))
    g._nnz = ea + eb
    g.static = a.static and b.static
    return g


def _kstep_plan(g, transposed):
    import ctypes
    L = _lib.lib()
    n = g.num_nodes
    if n == 0 or n > L.gda_kstep_max_rows():
        return None
    rp, ci, va = (g.t_rowptr, g.t_colidx, g.t_val) if transposed else (g.rowptr, g.colidx, g.val)
    nnz = g.nnz
    rp_h = rp.cpu().numpy()
    ci_h = ci[:nnz].cpu().numpy()
    va_h = va[:nnz].cpu().numpy()
    cap = L.gda_kstep_plan_bytes(12)
    buf = torch.empty(cap, dtype=torch.uint8)
    slots = L.gda_kstep_plan_host_ex(rp_h.ctypes.data, ci_h.ctypes.data if nnz else None,
                                     va_h.ctypes.data if nnz else None, n, KSTEP_BANK_AWARE, buf.data_ptr(), cap)
    if slots < 0:
        _lib.check(slots, "gda_kstep_plan_host_ex")
    if slots == 0:
        return None
    return buf[:L.gda_kstep_plan_bytes(slots)].to(g.device), int(slots)


KSTEP_LDS = os.environ.get("PYGDA_AMD_KSTEP_LDS", "1") == "1"
KSTEP_BANK_AWARE = int(os.environ.get("PYGDA_AMD_KSTEP_BANKS", "1"))     # bank-aware node placement in LDS (gda_kstep.hip)
KSTEP_LDS_MIN_K = int(os.environ.get("PYGDA_AMD_KSTEP_LDS_MIN_K", "3"))

SQUARE_MAX_FILL = float(os.environ.get("PYGDA_AMD_SQUARE_MAX_FILL", "6"))
SQUARE = os.environ.get("PYGDA_AMD_SQUARE", "0") == "1"      # opt-in: see ops.spmm_kstep


def _square_half(rowptr, colidx, val, n, nnz, max_nnz):
    import ctypes
    import numpy as np
    L = _lib.lib()
    rp = rowptr.cpu().numpy()
    ci = colidx[:nnz].cpu().numpy()
    va = val[:nnz].cpu().numpy()
    h = ctypes.c_void_p()
    _lib.check(L.gda_csr_square_host(rp.ctypes.data, ci.ctypes.data, va.ctypes.data, n,
                                     max(1, min(16, os.cpu_count() or 1)), int(max_nnz), ctypes.byref(h)),
               "gda_csr_square_host")
    try:
        m = L.gda_edge_list_size(h)
        if m == 0:
            return None
        ei = np.empty((2, m), dtype=np.int64)
        w = np.empty(m, dtype=np.float32)
        _lib.check(L.gda_edge_list_fetch(h, ei[0].ctypes.data, ei[1].ctypes.data, w.ctypes.data), "gda_edge_list_fetch")
    finally:
        L.gda_edge_list_destroy(h)
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(ei[0], minlength=n), out=ptr[1:])
    dev = rowptr.device
    return (torch.from_numpy(ptr.astype(np.int32)).to(dev), torch.from_numpy(ei[1].astype(np.int32)).to(dev),
            torch.from_numpy(w).to(dev))


def _square(g):
    nnz = g.nnz
    if nnz == 0:
        return None
    limit = int(SQUARE_MAX_FILL * nnz)
    fwd = _square_half(g.rowptr, g.colidx, g.val, g.num_nodes, nnz, limit)
    if fwd is None:
        return None
    bwd = _square_half(g.t_rowptr, g.t_colidx, g.t_val, g.num_nodes, nnz, -1)
    sq = CSRGraph(g.num_nodes, int(fwd[1].numel()), fwd[0], fwd[1], fwd[2], bwd[0], bwd[1], bwd[2])
    sq._nnz = int(fwd[1].numel())
    sq._squared = False
    return sq


def build_csr(edge_index, num_nodes, edge_weight=None, improved=False, add_self_loops=True,
              normalize=True, degree_side="col", validate=True, with_edge_map=False):
    """COO ``edge_index`` (row 0 = source, row 1 = destination) -> :class:`CSRGraph`.

    Semantics of gcn_norm (prop_gcn_conv.py:64-81; ``degree_side='col'``) and
    CachedGCNConv.norm (cached_gcn_conv.py:88-103; ``degree_side='row'``).
    ``add_self_loops='drop'`` removes existing self loops and appends none (the adjacency part of
    PyG ``get_laplacian``, dgsda_base.py:128).
    """
    _lib.require_gpu_tensor(edge_index, "edge_index", torch.int64)
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
    E, N = int(edge_index.size(1)), int(num_nodes)
    if validate and E > 0 and not getattr(edge_index, "_gda_trusted", False):   # sampler output: in range by construction
        lo, hi = int(edge_index.min()), int(edge_index.max())
        if lo < 0 or hi >= N:
            raise IndexError(f"edge_index values must lie in [0, {N}), found [{lo}, {hi}]")
    dev = edge_index.device
    src = edge_index[0].contiguous()
    dst = edge_index[1].contiguous()
    w = None
    if edge_weight is not None:
        _lib.require_gpu_tensor(edge_weight, "edge_weight")
        if edge_weight.numel() != E:
            raise ValueError("edge_weight must have one entry per edge")
        w = edge_weight.detach().to(torch.float32).contiguous()
    cap = E + N
    i32 = dict(dtype=torch.int32, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    rowptr, t_rowptr = torch.empty(N + 1, **i32), torch.empty(N + 1, **i32)
    colidx, t_colidx = torch.empty(max(cap, 1), **i32), torch.empty(max(cap, 1), **i32)
    val, t_val = torch.empty(max(cap, 1), **f32), torch.empty(max(cap, 1), **f32)
    L = _lib.lib()
    nbytes = L.gda_graph_workspace_bytes(E, N)
    ws = _lib.workspace(nbytes, dev, "graph")
    edge_map = torch.empty(max(cap, 1), **i32) if with_edge_map else None
    _lib.check(L.gda_build_csr_norm_map(
        _lib.ptr(src), _lib.ptr(dst), _lib.ptr(w), E, N, 2.0 if improved else 1.0,
        2 if add_self_loops == "drop" else int(bool(add_self_loops)), int(bool(normalize)),
        0 if degree_side == "col" else 1,
        _lib.ptr(rowptr), _lib.ptr(colidx), _lib.ptr(val),
        _lib.ptr(t_rowptr), _lib.ptr(t_colidx), _lib.ptr(t_val), _lib.ptr(edge_map),
        _lib.ptr(ws), ws.numel(), _lib.stream()), "gda_build_csr_norm_map")
    g = CSRGraph(N, cap, rowptr, colidx, val, t_rowptr, t_colidx, t_val)
    g.t_to_fwd = edge_map
    return g


class _GraphCache:
    """LRU of built graphs keyed by the identity (and in-place version) of the edge tensors."""

    def __init__(self, capacity=32):
        self.capacity = capacity
        self._d = OrderedDict()

    def get(self, edge_index, num_nodes, edge_weight, improved, add_self_loops, normalize, degree_side):
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes),
               None if edge_weight is None else (edge_weight.data_ptr(), edge_weight._version),
               bool(improved), add_self_loops if add_self_loops == "drop" else bool(add_self_loops),
               bool(normalize), degree_side)
        hit = self._d.get(key)
        if hit is not None:
            ref_ei, ref_w, g = hit
            if ref_ei() is edge_index and (edge_weight is None or ref_w() is edge_weight):
                self._d.move_to_end(key)
                return g
        g = build_csr(edge_index, num_nodes, edge_weight, improved, add_self_loops, normalize, degree_side)
        self._d[key] = (weakref.ref(edge_index), None if edge_weight is None else weakref.ref(edge_weight), g)
        while len(self._d) > self.capacity:
            self._d.popitem(last=False)
        return g

    def clear(self):
        self._d.clear()


graph_cache = _GraphCache()


def as_graph(edge_index, num_nodes, edge_weight=None, improved=False, add_self_loops=True,
             normalize=True, degree_side="col"):
    """``edge_index`` may already be a :class:`CSRGraph` (then it is used as is)."""
    if isinstance(edge_index, CSRGraph):
        return edge_index
    pre = getattr(edge_index, "_gda_prebuilt", None)   # a sampled batch whose gcn_norm graph the sampler built (sampler.py)
    if (pre is not None and edge_weight is None and not improved and (add_self_loops is True or add_self_loops == 1) and normalize
            and degree_side == "col" and pre.num_nodes == int(num_nodes)):
        sl = pre._slot
        if sl is not None and sl.gen != pre._gen:
            # the batch came out of a recycling loader ring (NeighborLoader(recycle=True)) and the ring has come round:
            # its block -- node ids, edge list, both CSRs, the K-step plans -- now holds ANOTHER batch's graph
            raise _lib.GdaError("this mini-batch's graph has been recycled: a batch of NeighborLoader(recycle=True) is "
                                "valid until prefetch + 4 further batches have been taken from the same loader "
                                "(keep a batch longer with recycle=False / PYGDA_AMD_LOADER_RECYCLE=0, or clone it)")
        return pre
    g = graph_cache.get(edge_index, num_nodes, edge_weight, improved, add_self_loops, normalize, degree_side)
    if getattr(edge_index, "_gda_static", False):      # tagged by the full-batch loader (pygda_amd/data.py)
        g.static = True
    if getattr(edge_index, "_gda_trusted", False):     # relabelled sub-graph of one mini-batch (pygda_amd/sampler.py)
        g.transient = True
    return g
