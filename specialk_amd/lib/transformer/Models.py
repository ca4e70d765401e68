"""The Transformer encoder-decoder with the reference's interface
(lib/transformer/Models.py:21-280): same constructors, attribute names, state_dict keys and
shapes, forward signatures and return tuples.  nn.Embedding / nn.Linear children hold the
parameters; arithmetic runs in HIP kernels (specialk_amd.ops).

Extension: every module carries `compute_dtype` (torch.float32 by default, the reference's
arithmetic; torch.bfloat16 for bf16 activations / MFMA with fp32 master weights), set for a
whole model with `set_compute_dtype`.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import _lib as _L
from ... import ops
from ...core import constants as Constants
from .Layers import DecoderLayer, EncoderLayer


def get_non_pad_mask(seq):
    """(B,L) ids -> (B,L,1) float, 1 on real tokens (lib/transformer/Models.py:21-23)."""
    assert seq.dim() == 2
    return seq.ne(Constants.PAD).type(torch.float).unsqueeze(-1)


def get_sinusoid_encoding_table(n_position, d_hid, padding_idx=None):
    """Sinusoid position encoding table (lib/transformer/Models.py:26-46): angles in float64,
    sin on even columns, cos on odd ones, row `padding_idx` zeroed, returned as float32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    hid = np.arange(d_hid)
    table = pos / np.power(10000, 2 * (hid // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    if padding_idx is not None:
        table[padding_idx] = 0.0
    return torch.FloatTensor(table)


def get_attn_key_pad_mask(seq_k, seq_q):
    """Dense (B,Lq,Lk) bool mask of PAD keys (lib/transformer/Models.py:49-57).  Kept for
    callers; the stacks below pass the (B,Lk) bitmap to the kernels instead."""
    len_q = seq_q.size(1)
    return seq_k.eq(Constants.PAD).unsqueeze(1).expand(-1, len_q, -1)


def get_subsequent_mask(seq):
    """Dense (B,L,L) uint8 strictly-upper-triangular mask (lib/transformer/Models.py:60-69)."""
    sz_b, len_s = seq.size()
    mask = torch.triu(torch.ones((len_s, len_s), device=seq.device, dtype=torch.uint8), diagonal=1)
    return mask.unsqueeze(0).expand(sz_b, -1, -1)


def _check_ids(seq, pos, n_vocab, n_pos):
    # nn.Embedding raises on out-of-range ids; the gather kernel would read out of bounds.  (One host read for the four
    # extremes.)
    if not seq.numel():
        return
    hi_s, lo_s, hi_p, lo_p = torch.stack((seq.max(), seq.min(), pos.max(), pos.min())).tolist()
    if hi_s >= n_vocab or lo_s < 0:
        raise IndexError(f"token id out of range [0, {n_vocab})")
    if hi_p >= n_pos or lo_p < 0:
        raise IndexError(f"position {hi_p} outside the position table of {n_pos} rows")


class PackPlan:
    """Row bookkeeping of a right-padded (B, L) id batch stored without its padding: sentence b
    owns rows off[b] .. off[b+1]-1 of every packed activation; idx[r] is the flat position
    b*L + i of packed row r inside the padded batch."""

    def __init__(self, off, idx, B, L, max_len, n):
        self.B, self.L, self.max_len, self.n = B, L, int(max_len), int(n)
        self.off = off
        self.idx = idx[: self.n]


class PendingPlans:
    """Packing plans whose host-side facts (row counts, longest sentence, validity flags) are still on their way:
    the device -> host copy was started without waiting.  `result()` waits for it (no wait at all if the copy was
    started a step earlier, see Transformer.plan_ahead) and returns the PackPlans."""

    def __init__(self, seqs, parts, facts_host, event, checked):
        self.seqs, self.parts, self.facts_host, self.event, self.checked = seqs, parts, facts_host, event, checked
        self._plans = None

    def matches(self, *seqs):
        return len(seqs) == len(self.seqs) and all(a is b for a, b in zip(seqs, self.seqs))

    def result(self):
        if self._plans is None:
            self.event.synchronize()
            facts = self.facts_host.tolist()
            plans = []
            for seq, (off, idx), (ok, lo, mx, n, ids_ok, pos_ok) in zip(self.seqs, self.parts, facts):
                if not ok or lo < 0:
                    raise ValueError("packed execution needs right-padded batches whose sentences hold at least `trim` tokens")
                if not (ids_ok and pos_ok):
                    raise IndexError("token id or position outside its embedding table")
                plans.append(PackPlan(off, idx, seq.shape[0], seq.shape[1], mx, n))
            self._plans = plans
        return self._plans


def plan_packing_async(seqs, trims, check=None):
    """Starts the packing plans of right-padded id batches `seqs` (each (B, L) on the GPU), keeping the first
    len_b - trim tokens of every sentence.  Lengths, offsets and row indices are computed on the device (two
    launches for all batches together); what the host must know (row counts, longest sentence, whether the batches
    really are right-padded and, if `check` is given — per batch (positions, vocabulary size, position-table rows) —
    whether ids / positions fit their tables) travels in ONE device -> host copy into pinned memory that is not
    waited for here."""
    items = []
    for k, (seq, trim) in enumerate(zip(seqs, trims)):
        pos, n_vocab, n_pos = check[k] if check is not None else (None, 0, 0)
        items.append((seq, pos, trim, n_vocab, n_pos))
    outs, facts = _L.pack_plans(items, Constants.PAD)
    parts = [(off, idx) for _, off, idx in outs]
    host = torch.empty(facts.shape, dtype=facts.dtype, pin_memory=True)
    host.copy_(facts, non_blocking=True)
    event = torch.cuda.Event()
    event.record()
    return PendingPlans(list(seqs), parts, host, event, check is not None)


def plan_packing(seqs, trims, check=None):
    """PackPlans for `seqs` (see plan_packing_async), waiting for the one device -> host copy."""
    return plan_packing_async(seqs, trims, check).result()


class Encoder(nn.Module):
    """A encoder model with self attention mechanism (lib/transformer/Models.py:72-129)."""

    def __init__(self, n_src_vocab, len_max_seq, d_word_vec, n_layers, n_head, d_k, d_v, d_model, d_inner,
                 dropout=0.1):
        super().__init__()
        n_position = len_max_seq + 1
        self.src_word_emb = nn.Embedding(n_src_vocab, d_word_vec, padding_idx=Constants.PAD)
        self.position_enc = nn.Embedding.from_pretrained(
            get_sinusoid_encoding_table(n_position, d_word_vec, padding_idx=0), freeze=True)
        self.layer_stack = nn.ModuleList(
            [EncoderLayer(d_model, d_inner, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers)])
        self.compute_dtype = torch.float32
        self.check_ids = True

    def forward(self, src_seq, src_pos, return_attns=False):
        B, Ls = src_seq.shape
        if self.check_ids:
            _check_ids(src_seq, src_pos, self.src_word_emb.num_embeddings, self.position_enc.num_embeddings)
        key_pad = src_seq.eq(Constants.PAD).to(torch.uint8).contiguous()  # (B, Ls); also the pad-row bitmap
        x = ops.call(ops.EmbedFn, src_seq, src_pos, self.src_word_emb.weight, self.position_enc.weight,
                              self.compute_dtype)
        attns = []
        for layer in self.layer_stack:
            x, attn = layer.forward_2d(x, B, Ls, key_pad=key_pad, row_pad=key_pad.view(-1), want_attn=return_attns)
            if return_attns:
                attns += [attn]
        enc_output = x.view(B, Ls, -1)
        if return_attns:
            return enc_output, attns
        return (enc_output,)


    def forward_packed(self, src_seq, src_pos, plan):
        """Encoder over the packed rows of `plan` (no padding rows anywhere): -> (n_src_tokens, d)."""
        flat = plan.idx
        x = ops.call(ops.EmbedFn, src_seq.reshape(-1)[flat], src_pos.reshape(-1)[flat], self.src_word_emb.weight,
                              self.position_enc.weight, self.compute_dtype)
        for layer in self.layer_stack:
            x, _ = layer.forward_2d(x, plan.B, plan.max_len, off=plan.off)
        return x


class Decoder(nn.Module):
    """A decoder model with self attention mechanism (lib/transformer/Models.py:132-199)."""

    def __init__(self, n_tgt_vocab, len_max_seq, d_word_vec, n_layers, n_head, d_k, d_v, d_model, d_inner,
                 dropout=0.1):
        super().__init__()
        n_position = len_max_seq + 1
        self.tgt_word_emb = nn.Embedding(n_tgt_vocab, d_word_vec, padding_idx=Constants.PAD)
        self.position_enc = nn.Embedding.from_pretrained(
            get_sinusoid_encoding_table(n_position, d_word_vec, padding_idx=0), freeze=True)
        self.layer_stack = nn.ModuleList(
            [DecoderLayer(d_model, d_inner, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers)])
        self.compute_dtype = torch.float32
        self.check_ids = True

    def forward(self, tgt_seq, tgt_pos, src_seq, enc_output, return_attns=False):
        B, Lt = tgt_seq.shape
        Ls = src_seq.shape[1]
        if self.check_ids:
            _check_ids(tgt_seq, tgt_pos, self.tgt_word_emb.num_embeddings, self.position_enc.num_embeddings)
        key_pad = tgt_seq.eq(Constants.PAD).to(torch.uint8).contiguous()
        src_key_pad = src_seq.eq(Constants.PAD).to(torch.uint8).contiguous()
        x = ops.call(ops.EmbedFn, tgt_seq, tgt_pos, self.tgt_word_emb.weight, self.position_enc.weight,
                              self.compute_dtype)
        enc2d = enc_output.reshape(B * Ls, -1)
        if enc2d.dtype != self.compute_dtype:
            enc2d = enc2d.to(self.compute_dtype)
        slf_attns, enc_attns = [], []
        # the k | v projections of the encoder output for all layers in one product
        kv_all, pres = ops.cross_kv(enc2d, [layer.enc_attn for layer in self.layer_stack])
        for li, layer in enumerate(self.layer_stack):
            x, a1, a2 = layer.forward_2d(x, enc2d if kv_all is None else kv_all, B, Lt, Ls, key_pad=key_pad,
                                         src_key_pad=src_key_pad, causal=True, row_pad=key_pad.view(-1),
                                         want_attn=return_attns, kv_pre=None if kv_all is None else pres[li])
            if return_attns:
                slf_attns += [a1]
                enc_attns += [a2]
        dec_output = x.view(B, Lt, -1)
        if return_attns:
            return dec_output, slf_attns, enc_attns
        return (dec_output,)


    def forward_packed(self, tgt_seq, tgt_pos, plan, enc_packed, src_plan):
        """Decoder over packed target rows attending to packed encoder rows: -> (n_tgt_tokens, d)."""
        flat = plan.idx
        x = ops.call(ops.EmbedFn, tgt_seq.reshape(-1)[flat], tgt_pos.reshape(-1)[flat], self.tgt_word_emb.weight,
                              self.position_enc.weight, self.compute_dtype)
        kv_all, pres = ops.cross_kv(enc_packed, [layer.enc_attn for layer in self.layer_stack])
        for li, layer in enumerate(self.layer_stack):
            x, _, _ = layer.forward_2d(x, enc_packed if kv_all is None else kv_all, plan.B, plan.max_len,
                                       src_plan.max_len, causal=True, off=plan.off, src_off=src_plan.off,
                                       kv_pre=None if kv_all is None else pres[li])
        return x


class Transformer(nn.Module):
    """A sequence to sequence model with attention mechanism (lib/transformer/Models.py:202-280)."""

    def __init__(self, n_src_vocab, n_tgt_vocab, len_max_seq, d_word_vec=512, d_model=512, d_inner=2048, n_layers=6,
                 n_head=8, d_k=64, d_v=64, dropout=0.1, tgt_emb_prj_weight_sharing=True,
                 emb_src_tgt_weight_sharing=True):
        super().__init__()
        self.encoder = Encoder(n_src_vocab=n_src_vocab, len_max_seq=len_max_seq, d_word_vec=d_word_vec,
                               d_model=d_model, d_inner=d_inner, n_layers=n_layers, n_head=n_head, d_k=d_k, d_v=d_v,
                               dropout=dropout)
        self.decoder = Decoder(n_tgt_vocab=n_tgt_vocab, len_max_seq=len_max_seq, d_word_vec=d_word_vec,
                               d_model=d_model, d_inner=d_inner, n_layers=n_layers, n_head=n_head, d_k=d_k, d_v=d_v,
                               dropout=dropout)
        self.generator = nn.Linear(d_model, n_tgt_vocab, bias=False)
        nn.init.xavier_normal_(self.generator.weight)
        assert d_model == d_word_vec, (
            "To facilitate the residual connections, the dimensions of all module outputs shall be the same "
            "(d_model must equal d_word_vec).")
        if tgt_emb_prj_weight_sharing:
            # target embedding and output projection share one matrix; logits are rescaled
            self.generator.weight = self.decoder.tgt_word_emb.weight
            self.x_logit_scale = d_model ** -0.5
        else:
            self.x_logit_scale = 1.0
        if emb_src_tgt_weight_sharing:
            assert n_src_vocab == n_tgt_vocab, (
                "To share word embedding table, the vocabulary size of src/tgt shall be the same.")
            self.encoder.src_word_emb.weight = self.decoder.tgt_word_emb.weight

    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.encoder.compute_dtype = dtype
        self.decoder.compute_dtype = dtype
        return self

    @property
    def compute_dtype(self):
        return self.decoder.compute_dtype

    def project(self, dec_output_2d, scaled=True):
        """generator(dec_output) [* x_logit_scale] on (T, d) rows."""
        return ops.linear(dec_output_2d, self.generator.weight, self.generator.bias,
                          alpha=self.x_logit_scale if scaled else 1.0)

    def forward(self, src_seq, src_pos, tgt_seq, tgt_pos):
        tgt_seq, tgt_pos = tgt_seq[:, :-1], tgt_pos[:, :-1]
        enc_output, *_ = self.encoder(src_seq, src_pos)
        dec_output, *_ = self.decoder(tgt_seq, tgt_pos, src_seq, enc_output)
        B, Lt, d = dec_output.shape
        seq_logit = self.project(dec_output.reshape(B * Lt, d))
        return seq_logit.view(B, Lt, -1)

    def plan_ahead(self, src_seq, src_pos, tgt_seq, tgt_pos):
        """Starts the packing plans of a batch for forward_packed(..., plans=) without waiting for the host copy.
        A training loop calls this for batch k + 1 before it runs batch k — as a loader prefetches — so that the
        host never waits for the device at a step boundary and keeps queueing work ahead of it."""
        check = None
        if self.encoder.check_ids:
            enc, dec = self.encoder, self.decoder
            check = [(src_pos, enc.src_word_emb.num_embeddings, enc.position_enc.num_embeddings),
                     (tgt_pos, dec.tgt_word_emb.num_embeddings, dec.position_enc.num_embeddings)]
        return plan_packing_async([src_seq, tgt_seq], [0, 1], check)

    def forward_packed(self, src_seq, src_pos, tgt_seq, tgt_pos, plans=None):
        """Training-time fast path: the same network on the batch WITHOUT its padding.

        Returns (logits (n, V), gold (n,)) for the n non-pad target positions, sentence after sentence —
        the rows of `forward(...)`'s result whose gold token is not PAD, in the same order, so
        the summed loss, the accuracy count and every parameter gradient equal the padded path's.
        Each sentence's last target token is never an input (forward drops it for the longest
        sentence; for shorter ones its row predicts PAD, is ignored by the loss and — the decoder
        being causal — seen by no other row), so packed target rows are tokens 0 .. len-2.
        Requires right-padded batches (what collate_fn builds); raises ValueError otherwise.
        `plans`: the result of plan_ahead() for these very tensors (ignored otherwise)."""
        if plans is None or not plans.matches(src_seq, tgt_seq):
            plans = self.plan_ahead(src_seq, src_pos, tgt_seq, tgt_pos)
        src_plan, tgt_plan = plans.result()
        enc_packed = self.encoder.forward_packed(src_seq, src_pos, src_plan)
        dec_packed = self.decoder.forward_packed(tgt_seq, tgt_pos, tgt_plan, enc_packed, src_plan)
        gold = tgt_seq.reshape(-1)[tgt_plan.idx + 1]
        return self.project(dec_packed), gold
