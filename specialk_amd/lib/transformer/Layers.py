"""Encoder / decoder layers with the reference's interface (lib/transformer/Layers.py:10-60).

`forward` accepts the reference's arguments (a (B,L,1) float non_pad_mask and dense bool
attention masks).  The encoder / decoder stacks call `forward_2d` instead, passing pad bitmaps,
so that no (B,Lq,Lk) mask tensor is built and the `*= non_pad_mask` after every sublayer is
folded into the LayerNorm kernel's epilogue.
"""
import torch
import torch.nn as nn

from .SubLayers import MultiHeadAttention, PositionwiseFeedForward


def _row_pad(non_pad_mask):
    if non_pad_mask is None:
        return None
    return non_pad_mask.reshape(-1).eq(0).to(torch.uint8)


def _dense(mask):
    return None if mask is None else mask.to(torch.uint8).contiguous()


class EncoderLayer(nn.Module):
    """Self-attention + feed-forward (lib/transformer/Layers.py:10-27)."""

    def __init__(self, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=dropout)

    def forward_2d(self, x, B, L, key_pad=None, dense_mask=None, row_pad=None, want_attn=False, off=None):
        x, attn = self.slf_attn.forward_2d(x, x, B, L, L, key_pad=key_pad, dense_mask=dense_mask, row_pad=row_pad,
                                           want_attn=want_attn, self_attn=True, q_off=off, k_off=off)
        x = self.pos_ffn.forward_2d(x, row_pad=row_pad)
        return x, attn

    def forward(self, enc_input, non_pad_mask=None, slf_attn_mask=None):
        B, L, d = enc_input.shape
        out, attn = self.forward_2d(enc_input.reshape(-1, d), B, L, dense_mask=_dense(slf_attn_mask),
                                    row_pad=_row_pad(non_pad_mask), want_attn=True)
        return out.view(B, L, d), attn


class DecoderLayer(nn.Module):
    """Masked self-attention + encoder-decoder attention + feed-forward
    (lib/transformer/Layers.py:30-60)."""

    def __init__(self, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.enc_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=dropout)

    def forward_2d(self, x, enc, B, Lt, Ls, key_pad=None, src_key_pad=None, causal=True, slf_dense=None,
                   cross_dense=None, row_pad=None, want_attn=False, off=None, src_off=None, kv_pre=None):
        x, a1 = self.slf_attn.forward_2d(x, x, B, Lt, Lt, key_pad=key_pad, causal=causal, dense_mask=slf_dense,
                                         row_pad=row_pad, want_attn=want_attn, self_attn=True, q_off=off, k_off=off)
        x, a2 = self.enc_attn.forward_2d(x, enc, B, Lt, Ls, key_pad=src_key_pad, dense_mask=cross_dense,
                                         row_pad=row_pad, want_attn=want_attn, self_attn=False, q_off=off,
                                         k_off=src_off, kv_pre=kv_pre)
        x = self.pos_ffn.forward_2d(x, row_pad=row_pad)
        return x, a1, a2

    def forward(self, dec_input, enc_output, non_pad_mask=None, slf_attn_mask=None, dec_enc_attn_mask=None):
        B, Lt, d = dec_input.shape
        Ls = enc_output.shape[1]
        out, a1, a2 = self.forward_2d(dec_input.reshape(-1, d), enc_output.reshape(-1, d), B, Lt, Ls, causal=False,
                                      slf_dense=_dense(slf_attn_mask), cross_dense=_dense(dec_enc_attn_mask),
                                      row_pad=_row_pad(non_pad_mask), want_attn=True)
        return out.view(B, Lt, d), a1, a2
