"""Multi-head attention and position-wise feed-forward sublayers.

Same constructors, parameter names/shapes (state_dict compatible) and forward signatures as
the reference's lib/transformer/SubLayers.py:12-85.  The nn.Linear / nn.Conv1d / nn.LayerNorm
children only hold (and initialise) the parameters; the arithmetic is done by HIP kernels via
specialk_amd.ops.  Each module also has a `forward_2d` used by the layer stack, which works on
flattened (B*L, d) activations and derives masks from pad bitmaps instead of dense tensors.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import ops
from .Modules import ScaledDotProductAttention


class MultiHeadAttention(nn.Module):
    """Multi-Head Attention module (lib/transformer/SubLayers.py:12-65)."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = nn.Linear(d_model, n_head * d_k)
        self.w_ks = nn.Linear(d_model, n_head * d_k)
        self.w_vs = nn.Linear(d_model, n_head * d_v)
        nn.init.normal_(self.w_qs.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_ks.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_vs.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_v)))
        self.attention = ScaledDotProductAttention(temperature=np.power(d_k, 0.5))
        self.layer_norm = nn.LayerNorm(d_model)
        self.fc = nn.Linear(n_head * d_v, d_model)
        nn.init.xavier_normal_(self.fc.weight)
        self.dropout = nn.Dropout(dropout)

    # -- fast path used by the encoder / decoder stacks
    def forward_2d(self, q_in, kv_in, B, Lq, Lk, key_pad=None, causal=False, dense_mask=None, row_pad=None,
                   want_attn=False, self_attn=None, q_off=None, k_off=None, kv_pre=None):
        """q_in (B*Lq, d), kv_in (B*Lk, d) -> (LayerNorm(fc(attn) + q_in) with pad rows zeroed, attn|None).
        With q_off / k_off (int32 [B+1] row offsets) the inputs are packed token rows and Lq / Lk the
        longest sequences.  kv_pre: kv_in holds the k | v projections of all decoder layers already
        (ops.cross_kv) and kv_pre names this layer's columns."""
        training = self.training
        cfg = dict(B=B, Lq=Lq, Lk=Lk, H=self.n_head, dk=self.d_k, dv=self.d_v, q_off=q_off, k_off=k_off,
                   self_attn=(q_in is kv_in) if self_attn is None else self_attn,
                   key_pad=key_pad, causal=bool(causal), dense_mask=dense_mask,
                   scale=1.0 / float(self.attention.temperature),
                   p_attn=self.attention.dropout.p if training else 0.0, want_attn=want_attn, kv_pre=kv_pre)
        p_out = self.dropout.p if training else 0.0
        if ops.mha_ln_fusable(q_in, self.fc.weight, self.n_head * self.d_v):
            # output projection, dropout, residual, LayerNorm and row mask in one launch
            cfg.update(ln_row_pad=row_pad, ln_eps=self.layer_norm.eps, ln_p=p_out)
            return ops.call(ops.MHALNFn, q_in, kv_in, self.w_qs.weight, self.w_qs.bias, self.w_ks.weight, self.w_ks.bias,
                            self.w_vs.weight, self.w_vs.bias, self.fc.weight, self.fc.bias, self.layer_norm.weight,
                            self.layer_norm.bias, cfg)
        out, attn = ops.call(ops.MHAFn, q_in, kv_in, self.w_qs.weight, self.w_qs.bias, self.w_ks.weight, self.w_ks.bias,
                                    self.w_vs.weight, self.w_vs.bias, self.fc.weight, self.fc.bias, cfg)
        y = ops.add_ln(out, q_in, self.layer_norm.weight, self.layer_norm.bias, row_pad, self.layer_norm.eps, p_out,
                       res_feeds_x=True)
        return y, attn

    # -- reference interface
    def forward(self, q, k, v, mask=None):
        """q (B,Lq,d), k, v (B,Lk,d), mask bool (B,Lq,Lk) -> (output (B,Lq,d), attn (h*B,Lq,Lk))."""
        sz_b, len_q, d = q.shape
        len_k = k.shape[1]
        dense = None if mask is None else mask.to(torch.uint8).contiguous()
        if k is v or (k.data_ptr() == v.data_ptr() and k.shape == v.shape):
            q2, kv2 = q.reshape(-1, d), k.reshape(-1, d)
            y, attn = self.forward_2d(q2, kv2, sz_b, len_q, len_k, dense_mask=dense, want_attn=True,
                                      self_attn=(q is k))
            return y.view(sz_b, len_q, d), attn
        # distinct key / value sources: project separately, then the generic attention module
        h, dk, dv = self.n_head, self.d_k, self.d_v
        Q = ops.linear(q.reshape(-1, d), self.w_qs.weight, self.w_qs.bias).view(sz_b, len_q, h, dk)
        K = ops.linear(k.reshape(-1, d), self.w_ks.weight, self.w_ks.bias).view(sz_b, len_k, h, dk)
        V = ops.linear(v.reshape(-1, d), self.w_vs.weight, self.w_vs.bias).view(sz_b, len_k, h, dv)
        Q = Q.permute(2, 0, 1, 3).reshape(-1, len_q, dk)
        K = K.permute(2, 0, 1, 3).reshape(-1, len_k, dk)
        V = V.permute(2, 0, 1, 3).reshape(-1, len_k, dv)
        mask_h = None if mask is None else mask.repeat(h, 1, 1)
        out, attn = self.attention(Q, K, V, mask=mask_h)
        out = out.view(h, sz_b, len_q, dv).permute(1, 2, 0, 3).reshape(sz_b * len_q, h * dv)
        out = ops.linear(out, self.fc.weight, self.fc.bias)
        y = ops.add_ln(out, q.reshape(-1, d), self.layer_norm.weight, self.layer_norm.bias, None, self.layer_norm.eps,
                       self.dropout.p if self.training else 0.0)
        return y.view(sz_b, len_q, d), attn


class PositionwiseFeedForward(nn.Module):
    """A two-feed-forward-layer module (lib/transformer/SubLayers.py:68-85).  w_1 / w_2 stay
    nn.Conv1d(k=1) so checkpoints keep their (out, in, 1) weight shapes."""

    def __init__(self, d_in, d_hid, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Conv1d(d_in, d_hid, 1)
        self.w_2 = nn.Conv1d(d_hid, d_in, 1)
        self.layer_norm = nn.LayerNorm(d_in)
        self.dropout = nn.Dropout(dropout)

    def forward_2d(self, x, row_pad=None):
        y = ops.call(ops.FFNFn, x, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias)
        return ops.add_ln(y, x, self.layer_norm.weight, self.layer_norm.bias, row_pad, self.layer_norm.eps,
                          self.dropout.p if self.training else 0.0, res_feeds_x=True)

    def forward(self, x):
        shape = x.shape
        return self.forward_2d(x.reshape(-1, shape[-1])).view(shape)
