"""Scaled dot-product attention module with the reference's interface
(lib/transformer/Modules.py:8-28), executed by the fused HIP attention kernel."""
import numpy as np
import torch.nn as nn

from ... import ops


class ScaledDotProductAttention(nn.Module):
    """softmax(mask(q k^T / temperature)) v, dropout on the probabilities.

    forward(q, k, v, mask=None) takes head-major batches q (n,Lq,dk), k (n,Lk,dk), v (n,Lk,dv)
    and a bool mask (n,Lq,Lk) with True = masked, and returns (output, attn) like the
    reference.  The (n,Lq,Lk) score matrix is never written to memory except as the returned
    `attn`."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)  # kept for its `p`; the kernel applies it
        self.softmax = nn.Softmax(dim=2)

    def forward(self, q, k, v, mask=None):
        p = self.dropout.p if self.training else 0.0
        return ops.SDPAFn.apply(q, k, v, mask, 1.0 / float(self.temperature), p)


def temperature_for(d_k):
    return np.power(d_k, 0.5)
