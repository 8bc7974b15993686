"""Sparse input features for the layer-0 projection.

The citation feature matrices are 0/1 bag-of-words stored dense (``x [N, 6775]``, ~1 % non
zero; pygda/datasets/citation.py:156-163).  The reference multiplies them as dense GEMMs --
75 % of an A2GNN step's FLOPs.  Here a feature matrix that is sparse enough is ingested
once into CSR (+ the transposed CSR for the weight gradient) by the graph-ingestion kernel,
and ``x @ W^T`` runs on the aggregation SpMM kernel with ``W^T`` as the gathered operand:
the same products in index order, the structural zeros skipped (same result up to fp32
summation order).  Registration happens where feature matrices enter the device
(``Data.to``); lookups are by tensor identity, so hidden activations never pay a check.
"""
import os
import weakref

import torch

from . import _lib, profiler
from .graph import build_csr

DENSITY_THRESHOLD = 0.10       # above this the dense MFMA GEMM wins
MIN_WIDTH = 256                # hidden-width inputs are never worth it

_registry = {}                 # data_ptr -> (weakref(tensor), version, shape, SparseFeatures | None)


class SparseFeatures:
    """CSR of X (rows = nodes, cols = feature ids) and of X^T, device resident."""

    def __init__(self, x):
        n, f = x.shape
        nz = x.nonzero(as_tuple=False)                     # row-major order: (node, feature)
        vals = x[nz[:, 0], nz[:, 1]].contiguous()
        # ingestion kernel: "edge" feature -> node, weight = value; no loops, no normalisation
        self.graph = build_csr(torch.stack([nz[:, 1], nz[:, 0]]), max(n, f), vals, add_self_loops=False,
                               normalize=False, validate=False, with_edge_map=True)
        self.n, self.f, self.nnz = n, f, int(nz.size(0))
        self._t_map = None

    @property
    def t_map(self):
        """by-feature CSR entry -> by-node CSR position (int64, for index_select), built on first use."""
        if self._t_map is None:
            self._t_map = self.graph.t_to_fwd[:self.nnz].to(torch.int64)
        return self._t_map

    def dropped_values(self, p):
        """Inverted dropout applied to the stored non-zeros only (zeros stay zero either way, so this
        is ``F.dropout(x, p)`` on the dense matrix in distribution): values for both CSRs."""
        keep = (torch.rand(self.nnz, device=self.graph.val.device) >= p).to(torch.float32) / (1.0 - p)
        val = self.graph.val[:self.nnz] * keep
        return val, val.index_select(0, self.t_map)


def maybe_register(x):
    """Called once per device feature matrix; decides dense vs sparse (one host sync)."""
    if (not torch.is_tensor(x) or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32
            or x.size(1) < MIN_WIDTH or x.requires_grad or not x.is_contiguous()):
        return None
    key = x.data_ptr()
    hit = _registry.get(key)
    if hit is not None and hit[0]() is x and hit[1] == x._version:
        return hit[3]
    density = float(torch.count_nonzero(x)) / max(x.numel(), 1)
    sf = SparseFeatures(x) if density <= DENSITY_THRESHOLD else None
    _registry[key] = (weakref.ref(x), x._version, tuple(x.shape), sf)
    if len(_registry) > 64:
        for k in [k for k, v in _registry.items() if v[0]() is None]:
            del _registry[k]
    return sf


def lookup(x):
    hit = _registry.get(x.data_ptr())
    if hit is None or hit[0]() is not x or hit[1] != x._version:
        return None
    return hit[3]


def _spmm(rowptr, colidx, val, n_rows, dense, out_rows, split, bias=None):
    import ctypes
    d = dense.size(1)
    y = torch.empty(out_rows, d, dtype=torch.float32, device=dense.device)
    sp = split.struct(d)                       # frequent words are hub rows of X^T
    L = _lib.lib()
    _lib.check(L.gda_spmm_csr_split_f32(_lib.ptr(rowptr), _lib.ptr(colidx), _lib.ptr(val), n_rows, d, 1,
                                        _lib.ptr(dense), d, _lib.ptr(y), d, None, _lib.ptr(bias),
                                        ctypes.byref(sp) if sp is not None else None, _lib.stream()),
               "gda_spmm_csr_split_f32")
    return y


class _SparseLinear(torch.autograd.Function):
    """``y = X W^T`` with X in CSR: forward gathers rows of W^T, backward is the transposed
    SpMM ``gW^T = X^T gy``.  X itself carries no gradient (it is the input data)."""

    @staticmethod
    def forward(ctx, weight, sf, vals=None, bias=None):
        # [F, h]: no copy once the weight is stored gather-major (_store_gather_major).  Sharing one transposed copy
        # between the source and the target branch (formed before the streams fork) was measured and dropped: a
        # kernel ahead of the fork costs the captured step 60-70 us (profiles/HISTORY.md 4.7)
        wt = weight.t().contiguous()
        g = sf.graph
        val, t_val = (g.val, g.t_val) if vals is None else vals
        with profiler.region(f"sparse_projection[{sf.f}x{weight.size(0)}]", 1,
                             sf.nnz * 8 + (sf.n + 1) * 4 + 4 * (wt.numel() + sf.n * weight.size(0)),
                             2 * sf.nnz * weight.size(0)):
            y = _spmm(g.rowptr, g.colidx, val, sf.n, wt, sf.n, g.split(False),
                      None if bias is None else bias.contiguous())
        ctx.sf, ctx.t_val, ctx.has_bias = sf, t_val, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        sf = ctx.sf
        g = sf.graph
        gy = gy.contiguous()
        with profiler.region(f"sparse_projection_bwd[{sf.f}x{gy.size(1)}]", 1,
                             sf.nnz * 8 + (sf.f + 1) * 4 + 4 * (gy.numel() + sf.f * gy.size(1)),
                             2 * sf.nnz * gy.size(1)):
            gwt = _spmm(g.t_rowptr, g.t_colidx, ctx.t_val, sf.f, gy, sf.f, g.split(True))   # [F, h]
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[3]:
            from .ops import colsum
            gb = colsum(gy)
        return gwt.t(), None, None, gb


class _SparseLinearT(torch.autograd.Function):
    """``(X W^T)^T`` as ``[out, round_up(n, 4)]``: the layer-0 projection of sparse features handed to the LDS-resident
    K-step kernel in that kernel's column-major layout (``gda_spmm_csr_tout_f32``) -- same sums as :class:`_SparseLinear`,
    no transpose launch between projection and aggregation.  Backward takes the column-major gradient."""

    @staticmethod
    def forward(ctx, weight, sf):
        wt = weight.t().contiguous()
        g = sf.graph
        h, n = weight.size(0), sf.n
        n_pad = (n + 3) // 4 * 4
        yT = torch.empty(h, n_pad, dtype=torch.float32, device=wt.device)
        if n_pad != n:
            yT[:, n:].zero_()
        L = _lib.lib()
        with profiler.region(f"sparse_projection[{sf.f}x{h}]", 1,
                             sf.nnz * 8 + (sf.n + 1) * 4 + 4 * (wt.numel() + sf.n * h), 2 * sf.nnz * h):
            _lib.check(L.gda_spmm_csr_tout_f32(_lib.ptr(g.rowptr), _lib.ptr(g.colidx), _lib.ptr(g.val), n, h,
                                               _lib.ptr(wt), h, _lib.ptr(yT), n_pad, None, _lib.stream()),
                       "gda_spmm_csr_tout_f32")
        ctx.sf = sf
        return yT

    @staticmethod
    def backward(ctx, gT):
        sf = ctx.sf
        g = sf.graph
        gT = gT.contiguous()
        h, n_pad = gT.shape
        gy = torch.empty(sf.n, h, dtype=torch.float32, device=gT.device)   # [n, h]: the transposed SpMM gathers whole rows
        _lib.check(_lib.lib().gda_transpose_f32(_lib.ptr(gT), n_pad, _lib.ptr(gy), h, h, sf.n, _lib.stream()),
                   "gda_transpose_f32")
        with profiler.region(f"sparse_projection_bwd[{sf.f}x{gy.size(1)}]", 1,
                             sf.nnz * 8 + (sf.f + 1) * 4 + 4 * (gy.numel() + sf.f * gy.size(1)),
                             2 * sf.nnz * gy.size(1)):
            gwt = _spmm(g.t_rowptr, g.t_colidx, g.t_val, sf.f, gy, sf.f, g.split(True))   # [F, h]
        return gwt.t(), None


def sparse_linear_colmajor(weight, sf):
    """``X W^T`` as a :class:`~pygda_amd.ops.ColMajor`, or None when the feature matrix has rows beyond the hub-row
    split threshold (those take the split kernel and a transpose)."""
    from .ops import ColMajor
    if sf.graph.split(False).struct(weight.size(0)) is not None:
        return None
    _store_gather_major(weight)
    return ColMajor(_SparseLinearT.apply(weight, sf), sf.n)


class _SparseMatmul(torch.autograd.Function):
    """``y = X W`` for a weight stored ``[in, out]`` (CachedGCNConv / PPMIConv, cached_gcn_conv.py:130): the weight
    IS the gathered operand -- no transposed copy either way; ``gW = X^T gy`` comes out in the weight's layout."""

    @staticmethod
    def forward(ctx, weight, sf):
        g = sf.graph
        w = weight.contiguous()
        with profiler.region(f"sparse_projection[{sf.f}x{w.size(1)}]", 1,
                             sf.nnz * 8 + (sf.n + 1) * 4 + 4 * (w.numel() + sf.n * w.size(1)), 2 * sf.nnz * w.size(1)):
            y = _spmm(g.rowptr, g.colidx, g.val, sf.n, w, sf.n, g.split(False))
        ctx.sf = sf
        return y

    @staticmethod
    def backward(ctx, gy):
        sf = ctx.sf
        g = sf.graph
        gy = gy.contiguous()
        with profiler.region(f"sparse_projection_bwd[{sf.f}x{gy.size(1)}]", 1,
                             sf.nnz * 8 + (sf.f + 1) * 4 + 4 * (gy.numel() + sf.f * gy.size(1)), 2 * sf.nnz * gy.size(1)):
            gw = _spmm(g.t_rowptr, g.t_colidx, g.t_val, sf.f, gy, sf.f, g.split(True))       # [F, h]
        return gw, None


def sparse_matmul(sf, weight):
    """``X @ weight`` with ``weight [in, out]``."""
    return _SparseMatmul.apply(weight, sf)


OWN_LAYOUT = os.environ.get("PYGDA_AMD_SPARSE_WT_LAYOUT", "1") == "1"


def _store_gather_major(weight):
    """Re-lay a leaf ``weight [out, in]`` out in ``[in, out]`` memory order -- same shape, same values, transposed
    strides -- the first time it meets sparse input features: ``W^T`` is the operand the SpMM gathers rows of, and
    the weight gradient ``X^T gy`` comes out ``[in, out]``.  With the reference's row-major storage every step paid
    a transposed copy of the 867 k-element layer-0 weight per branch (6-8 us each, the first kernel of the target
    branch) and a transposing copy of its gradient at the very end of the backward pass (9 us on the critical
    path); stored gather-major, ``weight.t()`` is contiguous and the gradient meets autograd's layout contract as
    it is.  Adam is elementwise, so the optimiser works on the raw memory (pygda_amd/optim.py); ``state_dict`` /
    ``load_state_dict`` / ``.weight[i, j]`` see the same ``[out, in]`` tensor as before."""
    if (OWN_LAYOUT and weight.dim() == 2 and weight.is_leaf and weight.is_contiguous() and weight.is_cuda
            and min(weight.shape) > 1 and not torch.cuda.is_current_stream_capturing()):
        with torch.no_grad():
            weight.data = weight.data.t().contiguous().t()
            if weight.grad is not None:
                weight.grad = None


def sparse_linear(weight, sf, dropout=0.0, bias=None):
    """``X W^T (+ bias)``; with ``dropout > 0`` the product uses ``dropout(X)`` (mask drawn per call)."""
    _store_gather_major(weight)
    if dropout > 0.0:
        return _SparseLinear.apply(weight, sf, sf.dropped_values(dropout), bias)
    return _SparseLinear.apply(weight, sf, None, bias)
