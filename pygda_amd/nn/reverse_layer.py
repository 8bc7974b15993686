"""Gradient reversal (pygda/nn/reverse_layer.py:4-66): identity forward, ``-alpha * g``
backward.  Stand-alone form for user code; the trainers use the fused
GRL + discriminator + cross-entropy kernel (:func:`pygda_amd.ops.grl_disc_ce`)."""
import torch


class GradReverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.neg() * ctx.alpha, None
