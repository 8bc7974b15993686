"""View attention (pygda/nn/attention.py:6-55): softmax over K stacked views of a learned
per-view score, weighted sum.  K = 2 (GCN view, PPMI view) in UDAGCN.  Device views of up to 512 columns run on the fused
kernels of csrc/gda_attention.hip (no [N, K, h] temporaries); anything else composes the reference's lines."""
import torch
import torch.nn.functional as F
from torch import nn
from .linear import DenseLinear


class Attention(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.dense_weight = DenseLinear(in_channels, 1)
        self.dropout = nn.Dropout(0.1)     # constructed, never applied -- as in the reference (:26)

    def forward(self, inputs):
        from ..ops import attention_fuse, attention_fuse_ok
        if attention_fuse_ok(inputs, self.dense_weight.weight):      # device views: one fused pass each way (csrc/gda_attention.hip)
            return attention_fuse(list(inputs), self.dense_weight.weight, self.dense_weight.bias)
        stacked = torch.stack(inputs, dim=1)
        weights = F.softmax(self.dense_weight(stacked), dim=1)
        return torch.sum(stacked * weights, dim=1)
