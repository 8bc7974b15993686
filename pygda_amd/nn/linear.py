"""PyG-style ``Linear`` (weight ``[out, in]``, glorot) used inside the conv layers.

The dense hidden x weight contraction of a node matrix (1024 rows and more, weight extents up to 256) runs on the
hand-written fp32 matrix-core kernels of csrc/gda_gemm.hip -- 64 x 64 tiles at citation size, the weight-in-registers
kernels for sampled sub-graphs of 10^5 rows and more; other shapes (tiny row counts, wide weights) go through
``F.linear``.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import profiler, sparse_features


def glorot(t):
    """U(-a, a), a = sqrt(6 / (fan_in + fan_out)) -- PyG ``inits.glorot``."""
    if t is not None:
        a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
        with torch.no_grad():
            t.uniform_(-a, a)


def zeros(t):
    if t is not None:
        with torch.no_grad():
            t.fill_(0)


class _TallLinear(torch.autograd.Function):
    """``y = x W^T`` for a tall ``x [N, in]`` and a small ``W [out, in]`` (the hidden and classifier
    layers: N = nodes, in/out <= 256) on the hand-written matrix-core kernels (csrc/gda_gemm.hip):
    64x64 tiles fill the chip where the BLAS's 128x128 macro-tiles leave two thirds of it idle, and
    the weight gradient ``gW = gy^T x`` (tiny output, reduction over N) is a deterministic split
    over row slabs."""

    @staticmethod
    def forward(ctx, x, weight):
        from ..ops import GEMM_NT, gemm
        ctx.save_for_backward(x, weight)
        return gemm(GEMM_NT, x, weight)

    @staticmethod
    def backward(ctx, gy):
        from ..ops import GEMM_NN, GEMM_TN, gemm
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gemm(GEMM_NN, gy, weight) if ctx.needs_input_grad[0] else None
        gw = gemm(GEMM_TN, gy, x) if ctx.needs_input_grad[1] else None
        return gx, gw


class _TallLinearBias(torch.autograd.Function):
    """``y = x W^T + b`` (a conv with ``prop_nums = 0``, prop_gcn_conv.py:204,212-213): the bias rides in the forward
    kernel's epilogue and its gradient ``gy.sum(0)`` is a by-product of the weight-gradient kernel (column sums of
    the ``gy`` tiles it stages anyway) -- no elementwise add, no fill + reduction launches."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from ..ops import GEMM_NT, gemm
        ctx.save_for_backward(x, weight)
        return gemm(GEMM_NT, x, weight, bias=bias.contiguous())

    @staticmethod
    def backward(ctx, gy):
        from ..ops import GEMM_NN, GEMM_TN, gemm
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gemm(GEMM_NN, gy, weight) if ctx.needs_input_grad[0] else None
        gb = torch.empty(weight.size(0), dtype=torch.float32, device=gy.device)
        gw = gemm(GEMM_TN, gy, x, colsum=gb)
        return gx, gw, gb


def tall_linear_bias(x, weight, bias):
    return _TallLinearBias.apply(x, weight, bias)


class _TallMatmul(torch.autograd.Function):
    """``y = x W`` for a weight stored ``[in, out]`` (CachedGCNConv, cached_gcn_conv.py:130) on the same kernels:
    forward NN, data gradient ``gy W^T`` = NT, weight gradient ``x^T gy`` = TN in the weight's own layout."""

    @staticmethod
    def forward(ctx, x, weight):
        from ..ops import GEMM_NN, gemm
        ctx.save_for_backward(x, weight)
        return gemm(GEMM_NN, x, weight)

    @staticmethod
    def backward(ctx, gy):
        from ..ops import GEMM_NT, GEMM_TN, gemm
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gemm(GEMM_NT, gy, weight) if ctx.needs_input_grad[0] else None
        gw = gemm(GEMM_TN, x, gy) if ctx.needs_input_grad[1] else None
        return gx, gw


def tall_matmul_ok(x, weight):
    return (x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and 1024 <= x.size(0) and weight.size(0) <= 256 and weight.size(1) <= 256)


def tall_matmul(x, weight):
    return _TallMatmul.apply(x, weight)


class Linear(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, weight_initializer="glorot"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer = weight_initializer
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        """PyG ``Linear.reset_parameters()`` (dense/linear.py): 'glorot', or its defaults -- ``kaiming_uniform(weight,
        fan=in, a=sqrt(5))`` = U(+-sqrt(6 / ((1 + a^2) in))) and ``inits.uniform(in, bias)`` = U(+-1/sqrt(in)) --
        weight first, then bias: the order the init RNG stream is consumed in (tests/golden/_pyg_stub.py, 13a)."""
        if self.weight_initializer == "glorot":
            glorot(self.weight)
        else:
            bound = math.sqrt(6 / ((1 + math.sqrt(5) ** 2) * self.in_channels))
            with torch.no_grad():
                self.weight.uniform_(-bound, bound)
        if self.bias is not None:
            if self.weight_initializer == "glorot":
                zeros(self.bias)
            else:
                b = 1.0 / math.sqrt(self.in_channels)
                with torch.no_grad():
                    self.bias.uniform_(-b, b)

    def tall_gemm_ok(self, x):
        """``x`` goes to the hand-written matrix-core kernels (not the sparse-input path, not the BLAS)."""
        if self.bias is not None or x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32:
            return False
        if x.size(1) >= sparse_features.MIN_WIDTH and sparse_features.lookup(x) is not None:
            return False
        return 1024 <= x.size(0) and self.in_channels <= 256 and self.out_channels <= 256

    def forward(self, x):
        if self.bias is None and x.dim() == 2 and x.size(1) >= sparse_features.MIN_WIDTH:
            sf = sparse_features.lookup(x)             # identity lookup: input feature matrices only
            if sf is not None:
                return sparse_features.sparse_linear(self.weight, sf)
        if (self.bias is None and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and x.size(0) >= 1024
                and self.in_channels <= 256 and self.out_channels <= 256):
            return _TallLinear.apply(x, self.weight)          # every row count: ops.gemm picks the kernel by shape
        if profiler.enabled:
            n = x.numel() // x.size(-1)
            with profiler.region(f"dense_projection[{self.in_channels}x{self.out_channels}]", 1,
                                 4 * (x.numel() + self.weight.numel() + n * self.out_channels),
                                 2 * n * self.in_channels * self.out_channels):
                return F.linear(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, bias={self.bias is not None})"
