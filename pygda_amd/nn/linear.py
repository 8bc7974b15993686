"""PyG-style ``Linear`` (weight ``[out, in]``, glorot) used inside the conv layers.

The dense hidden x weight contraction of a node matrix runs on the hand-written fp32 matrix-core kernels of
csrc/gda_gemm.hip -- 64 x 64 tiles at citation size (any shape: gda_gemm_ex_f32), the weight-in-registers kernels for
sampled sub-graphs of 10^5 rows and more, the vector kernels for the classifier's few columns.  ``F.linear`` is left to
host tensors.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import sparse_features


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
        from ..ops import GEMM_NT, gemm, sink_of
        ctx.sink = sink_of(x)             # x = the output of dropout(relu(.)) that offered a GradSink (ops.grad_sinks)
        ctx.save_for_backward(x, weight)
        return gemm(GEMM_NT, x, weight)

    @staticmethod
    def backward(ctx, gy):
        from ..ops import GEMM_NN, GEMM_TN, gemm, masked_dgrad
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:       # (the data gradient stored through the activation's backward when it has a sink)
            gx = masked_dgrad(gy, weight, ctx.sink) if ctx.sink is not None else gemm(GEMM_NN, gy, weight)
        gw = gemm(GEMM_TN, gy, x) if ctx.needs_input_grad[1] else None
        return gx, gw


class _TallLinearBias(torch.autograd.Function):
    """``y = x W^T + b`` (a conv with ``prop_nums = 0``, prop_gcn_conv.py:204,212-213): the bias rides in the forward
    kernel's epilogue and its gradient ``gy.sum(0)`` is a by-product of the weight-gradient kernel (column sums of
    the ``gy`` tiles it stages anyway) -- no elementwise add, no fill + reduction launches."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from ..ops import GEMM_NT, gemm, sink_of
        ctx.sink = sink_of(x)
        ctx.save_for_backward(x, weight)
        return gemm(GEMM_NT, x, weight, bias=bias.contiguous())

    @staticmethod
    def backward(ctx, gy):
        from ..ops import GEMM_NN, GEMM_TN, gemm, masked_dgrad
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = masked_dgrad(gy, weight, ctx.sink) if ctx.sink is not None else gemm(GEMM_NN, gy, weight)
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


class DenseLinear(nn.Linear):
    """``torch.nn.Linear`` -- same parameters, same initialisation stream, same state dict -- whose product with a GPU
    tensor runs on the hand-written matrix-core kernels (bias in the forward epilogue, its gradient beside the weight
    gradient): the classifier heads and discriminators the reference builds from ``nn.Linear`` (a2gnn_base.py:62-66,
    grade_base.py:66-70, udagcn_base.py:100-108 ...).  NOT for modules that are differentiated twice (the gradient-penalty
    critics of specreg.py / adagcn.py keep ``nn.Linear``: this function's backward is not itself differentiable)."""

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32 and x.dim() >= 1:
            rows = x.reshape(-1, x.size(-1))
            y = (_TallLinear.apply(rows, self.weight) if self.bias is None
                 else _TallLinearBias.apply(rows, self.weight, self.bias))
            return y if x.dim() == 2 else y.view(*x.shape[:-1], self.out_features)
        return F.linear(x, self.weight, self.bias)


def tall_matmul_ok(x, weight):
    """Every 2-D fp32 product on the GPU (round 5: the general entry takes any shape; the BLAS is left to host tensors)."""
    return x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32


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
        if x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32 and x.dim() >= 1:
            # every other GPU shape (a bias, fewer than 1024 rows, extents above 256, batched inputs): the same hand-written
            # kernels through their general 64 x 64-tile entry (gda_gemm_ex_f32 takes any shape) -- no BLAS on the path
            rows = x.reshape(-1, x.size(-1))
            y = _TallLinear.apply(rows, self.weight) if self.bias is None else _TallLinearBias.apply(rows, self.weight, self.bias)
            return y if x.dim() == 2 else y.view(*x.shape[:-1], self.out_channels)
        return F.linear(x, self.weight, self.bias)              # host tensors (CPU-side tests of the module logic)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, bias={self.bias is not None})"
