#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the reference itself.

Runs ONLY in the build container (it imports /root/reference/base/lib/transformer, which
does not exist on the GPU box).  The fixtures it writes are plain .npz data (inputs, weights,
expected outputs), loadable with numpy.load(allow_pickle=False).

What is taken from the reference by import (unmodified):
  lib.transformer.Modules / SubLayers / Layers / Models / Translator / Beam / Optim
What cannot be imported here (lib.TransformerModel needs toolz/nltk/telebot) and is
therefore restated from its source text:
  calculate_loss / performance  (base/lib/TransformerModel.py:340-375)
  setup_optimiser               (base/lib/TransformerModel.py:155-168)
  initiate's xavier pass        (base/lib/TransformerModel.py:73-75)
Known defect worked around: Beam.advance (base/lib/transformer/Beam.py:70) divides an int64
tensor by an int expecting floor division (torch < 1.5).  We restore that semantics for the
duration of the call; no other line of the reference is changed.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import os
import sys
from argparse import Namespace

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/base"
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import torch.optim as optim  # noqa: E402

import lib.transformer.Translator as RefTranslatorMod  # noqa: E402
from lib.transformer.Beam import Beam as RefBeam  # noqa: E402
from lib.transformer.Layers import DecoderLayer, EncoderLayer  # noqa: E402
from lib.transformer.Models import Transformer, get_sinusoid_encoding_table  # noqa: E402
from lib.transformer.Modules import ScaledDotProductAttention  # noqa: E402
from lib.transformer.Optim import ScheduledOptim  # noqa: E402
from lib.transformer.SubLayers import MultiHeadAttention, PositionwiseFeedForward  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
PAD, UNK, SOS, EOS = 0, 1, 2, 3


class FloorDivBeam(RefBeam):
    """Reference Beam with `int_tensor / int` meaning floor division, as in torch < 1.5."""

    def advance(self, word_prob):
        orig = torch.Tensor.__truediv__

        def floordiv(a, b):
            if isinstance(a, torch.Tensor) and not a.is_floating_point() and isinstance(b, int):
                return torch.div(a, b, rounding_mode="floor")
            return orig(a, b)

        torch.Tensor.__truediv__ = floordiv
        try:
            return super().advance(word_prob)
        finally:
            torch.Tensor.__truediv__ = orig


RefTranslatorMod.Beam = FloorDivBeam


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def sd_arrays(prefix, module):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def grad_arrays(prefix, module):
    out = {}
    for k, p in module.named_parameters():
        if p.grad is not None:
            out[prefix + k] = npy(p.grad)
    return out


def xavier_like_initiate(model):
    # base/lib/TransformerModel.py:73-75
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


def randomize_small_params(module, g):
    """Biases / LayerNorm affine default to 0 / 1, which would hide indexing bugs."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))


def make_batch(g, n, lo, hi, vocab):
    """Sentences [SOS] ids [EOS], right-padded with PAD; pos = 1..len, 0 on pads
    (base/core/dataset.py:21-46)."""
    lens = torch.randint(lo, hi + 1, (n,), generator=g).tolist()
    L = max(lens) + 2
    seq = torch.zeros(n, L, dtype=torch.long)
    pos = torch.zeros(n, L, dtype=torch.long)
    for i, l in enumerate(lens):
        ids = torch.randint(5, vocab, (l,), generator=g)
        seq[i, 0] = SOS
        seq[i, 1 : l + 1] = ids
        seq[i, l + 1] = EOS
        pos[i, : l + 2] = torch.arange(1, l + 3)
    return seq, pos


def calculate_loss(pred, gold, smoothing):
    # restated from base/lib/TransformerModel.py:353-375
    gold = gold.contiguous().view(-1)
    if smoothing:
        epsilon = 0.1
        n_class = pred.size(1)
        one_hot = torch.zeros_like(pred).scatter(1, gold.view(-1, 1), 1)
        one_hot = one_hot * (1 - epsilon) + (1 - one_hot) * epsilon / (n_class - 1)
        log_prb = F.log_softmax(pred, dim=1)
        non_pad_mask = gold.ne(PAD)
        loss = -(one_hot * log_prb).sum(dim=1)
        loss = loss.masked_select(non_pad_mask).sum()
    else:
        loss = F.cross_entropy(pred, gold, ignore_index=PAD, reduction="sum")
    return loss


def performance(pred, gold, smoothing):
    # restated from base/lib/TransformerModel.py:340-351
    loss = calculate_loss(pred, gold, smoothing)
    pred = pred.max(1)[1]
    gold = gold.contiguous().view(-1)
    non_pad_mask = gold.ne(PAD)
    n_correct = pred.eq(gold).masked_select(non_pad_mask).sum().item()
    return loss, n_correct


# --------------------------------------------------------------------------------------
def gen_attention():
    g = torch.Generator().manual_seed(11)
    n_head, B, Lq, Lk, dk, dv = 2, 3, 6, 7, 8, 8
    att = ScaledDotProductAttention(temperature=float(np.power(dk, 0.5)))
    att.eval()
    arrs = {}
    for tag, (lq, lk, causal) in {"keypad": (Lq, Lq, False), "causal": (Lq, Lq, True), "cross": (Lq, Lk, False)}.items():
        q = torch.randn(n_head * B, lq, dk, generator=g, requires_grad=True)
        k = torch.randn(n_head * B, lk, dk, generator=g, requires_grad=True)
        v = torch.randn(n_head * B, lk, dv, generator=g, requires_grad=True)
        klen = torch.tensor([lk, lk - 2, lk - 3])
        keypad = torch.arange(lk)[None, :] >= klen[:, None]  # (B, lk) True = PAD
        mask = keypad[:, None, :].expand(B, lq, lk)
        if causal:
            sub = torch.triu(torch.ones(lq, lq, dtype=torch.uint8), diagonal=1)[None].expand(B, -1, -1)
            mask = (mask + sub).gt(0)
        mask_h = mask.repeat(n_head, 1, 1)
        out, attn = att(q, k, v, mask=mask_h)
        dout = torch.randn(out.shape, generator=g)
        out.backward(dout)
        arrs.update({
            f"{tag}_q": npy(q), f"{tag}_k": npy(k), f"{tag}_v": npy(v), f"{tag}_keypad": npy(keypad),
            f"{tag}_mask": npy(mask_h), f"{tag}_out": npy(out), f"{tag}_attn": npy(attn),
            f"{tag}_dout": npy(dout), f"{tag}_dq": npy(q.grad), f"{tag}_dk": npy(k.grad), f"{tag}_dv": npy(v.grad),
        })
    arrs["n_head"] = np.int64(n_head)
    save("attention", **arrs)


def gen_sublayers():
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(12)
    d_model, d_inner, n_head, dk, dv, B, Lq, Lk = 32, 64, 4, 8, 8, 3, 6, 7
    # --- MultiHeadAttention, self (causal) and cross
    for tag, lk, causal in (("mha_self", Lq, True), ("mha_cross", Lk, False)):
        mha = MultiHeadAttention(n_head, d_model, dk, dv, dropout=0.1)
        randomize_small_params(mha, g)
        mha.eval()
        x = torch.randn(B, Lq, d_model, generator=g, requires_grad=True)
        mem = x if tag == "mha_self" else torch.randn(B, lk, d_model, generator=g, requires_grad=True)
        klen = torch.tensor([lk, lk - 1, lk - 3])
        keypad = torch.arange(lk)[None, :] >= klen[:, None]
        mask = keypad[:, None, :].expand(B, Lq, lk)
        if causal:
            sub = torch.triu(torch.ones(Lq, Lq, dtype=torch.uint8), diagonal=1)[None].expand(B, -1, -1)
            mask = (mask + sub).gt(0)
        out, attn = mha(x, mem, mem, mask=mask)
        dout = torch.randn(out.shape, generator=g)
        out.backward(dout)
        arrs = {"x": npy(x), "mem": npy(mem), "keypad": npy(keypad), "mask": npy(mask), "causal": np.int64(causal),
                "out": npy(out), "attn": npy(attn), "dout": npy(dout), "dx": npy(x.grad)}
        if tag == "mha_cross":
            arrs["dmem"] = npy(mem.grad)
        arrs.update(sd_arrays("w:", mha))
        arrs.update(grad_arrays("g:", mha))
        arrs["cfg"] = np.array([n_head, d_model, dk, dv], dtype=np.int64)
        save(tag, **arrs)
    # --- PositionwiseFeedForward
    ffn = PositionwiseFeedForward(d_model, d_inner, dropout=0.1)
    randomize_small_params(ffn, g)
    ffn.eval()
    x = torch.randn(B, Lq, d_model, generator=g, requires_grad=True)
    out = ffn(x)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    arrs = {"x": npy(x), "out": npy(out), "dout": npy(dout), "dx": npy(x.grad)}
    arrs.update(sd_arrays("w:", ffn))
    arrs.update(grad_arrays("g:", ffn))
    save("ffn", **arrs)
    # --- EncoderLayer / DecoderLayer
    seq_len = torch.tensor([Lq, Lq - 2, Lq - 3])
    src_len = torch.tensor([Lk, Lk - 1, Lk - 4])
    non_pad = (torch.arange(Lq)[None, :] < seq_len[:, None]).float().unsqueeze(-1)
    keypad = torch.arange(Lq)[None, :] >= seq_len[:, None]
    src_keypad = torch.arange(Lk)[None, :] >= src_len[:, None]

    enc = EncoderLayer(d_model, d_inner, n_head, dk, dv, dropout=0.1)
    randomize_small_params(enc, g)
    enc.eval()
    x = torch.randn(B, Lq, d_model, generator=g, requires_grad=True)
    out, attn = enc(x, non_pad_mask=non_pad, slf_attn_mask=keypad[:, None, :].expand(B, Lq, Lq))
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    arrs = {"x": npy(x), "keypad": npy(keypad), "out": npy(out), "attn": npy(attn), "dout": npy(dout), "dx": npy(x.grad)}
    arrs.update(sd_arrays("w:", enc))
    arrs.update(grad_arrays("g:", enc))
    arrs["cfg"] = np.array([n_head, d_model, dk, dv, d_inner], dtype=np.int64)
    save("enc_layer", **arrs)

    dec = DecoderLayer(d_model, d_inner, n_head, dk, dv, dropout=0.1)
    randomize_small_params(dec, g)
    dec.eval()
    x = torch.randn(B, Lq, d_model, generator=g, requires_grad=True)
    mem = torch.randn(B, Lk, d_model, generator=g, requires_grad=True)
    sub = torch.triu(torch.ones(Lq, Lq, dtype=torch.uint8), diagonal=1)[None].expand(B, -1, -1)
    slf_mask = (keypad[:, None, :].expand(B, Lq, Lq) + sub).gt(0)
    out, a1, a2 = dec(x, mem, non_pad_mask=non_pad, slf_attn_mask=slf_mask,
                      dec_enc_attn_mask=src_keypad[:, None, :].expand(B, Lq, Lk))
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    arrs = {"x": npy(x), "mem": npy(mem), "keypad": npy(keypad), "src_keypad": npy(src_keypad), "out": npy(out),
            "slf_attn": npy(a1), "enc_attn": npy(a2), "dout": npy(dout), "dx": npy(x.grad), "dmem": npy(mem.grad)}
    arrs.update(sd_arrays("w:", dec))
    arrs.update(grad_arrays("g:", dec))
    arrs["cfg"] = np.array([n_head, d_model, dk, dv, d_inner], dtype=np.int64)
    save("dec_layer", **arrs)


def gen_sinusoid():
    save("sinusoid",
         t_20_32=npy(get_sinusoid_encoding_table(20, 32, padding_idx=0)),
         t_7_6=npy(get_sinusoid_encoding_table(7, 6, padding_idx=0)),
         t_5_8_nopad=npy(get_sinusoid_encoding_table(5, 8)))


TINY = dict(len_max_seq=12, d_word_vec=32, d_model=32, d_inner=64, n_layers=2, n_head=4, d_k=8, d_v=8, dropout=0.1)


def gen_transformer():
    for tag, tied, vocab in (("transformer_untied", False, (37, 41)), ("transformer_tied", True, (37, 37))):
        torch.manual_seed(21)
        g = torch.Generator().manual_seed(22)
        m = Transformer(vocab[0], vocab[1], tgt_emb_prj_weight_sharing=tied, emb_src_tgt_weight_sharing=tied, **TINY)
        xavier_like_initiate(m)
        randomize_small_params(m, g)
        m.eval()
        src_seq, src_pos = make_batch(g, 4, 2, 8, vocab[0])
        tgt_seq, tgt_pos = make_batch(g, 4, 2, 8, vocab[1])
        arrs = {"src_seq": npy(src_seq), "src_pos": npy(src_pos), "tgt_seq": npy(tgt_seq), "tgt_pos": npy(tgt_pos),
                "tied": np.int64(tied), "vocab": np.array(vocab, dtype=np.int64)}
        arrs.update(sd_arrays("w:", m))
        enc_out, = m.encoder(src_seq, src_pos)
        arrs["enc_out"] = npy(enc_out)
        for smoothing in (True, False):
            m.zero_grad()
            logits = m(src_seq, src_pos, tgt_seq, tgt_pos)
            gold = tgt_seq[:, 1:]
            loss, n_correct = performance(logits.view(-1, logits.size(2)), gold, smoothing)
            loss.backward()
            s = "ls" if smoothing else "ce"
            arrs["logits"] = npy(logits)
            arrs[f"loss_{s}"] = npy(loss)
            arrs[f"ncorrect_{s}"] = np.int64(n_correct)
            arrs.update(grad_arrays(f"g_{s}:", m))
        arrs["x_logit_scale"] = np.float64(m.x_logit_scale)
        save(tag, **arrs)


def gen_loss():
    g = torch.Generator().manual_seed(31)
    T, V = 11, 23
    pred = (3.0 * torch.randn(T, V, generator=g)).requires_grad_(True)
    gold = torch.randint(1, V, (T,), generator=g)
    gold[2] = PAD
    gold[7] = PAD
    gold[10] = PAD
    with torch.no_grad():
        pred[4, gold[4]] += 9.0   # a few guaranteed-correct rows
        pred[5, gold[5]] += 9.0
        pred[7, 3] += 9.0         # argmax on a PAD row must not count
    arrs = {"pred": npy(pred), "gold": npy(gold)}
    for smoothing in (True, False):
        pred.grad = None
        loss, nc = performance(pred, gold, smoothing)
        loss.backward()
        s = "ls" if smoothing else "ce"
        arrs[f"loss_{s}"] = npy(loss)
        arrs[f"dpred_{s}"] = npy(pred.grad)
        arrs[f"ncorrect_{s}"] = np.int64(nc)
    save("loss", **arrs)


def gen_optim():
    # ScheduledOptim (base/lib/transformer/Optim.py) over Adam as set up in
    # base/lib/TransformerModel.py:155-168
    g = torch.Generator().manual_seed(41)
    d_model, warm = 32, 4
    p1 = nn.Parameter(torch.randn(5, 7, generator=g))
    p2 = nn.Parameter(torch.randn(9, generator=g))
    frozen = nn.Parameter(torch.randn(3, generator=g), requires_grad=False)
    params = [p1, p2, frozen]
    so = ScheduledOptim(optim.Adam(filter(lambda x: x.requires_grad, params), betas=(0.9, 0.98), eps=1e-09, lr=1e-3),
                        d_model, warm)
    arrs = {"p1_0": npy(p1), "p2_0": npy(p2), "d_model": np.int64(d_model), "n_warmup": np.int64(warm)}
    lrs = []
    for step in range(1, 7):
        so.zero_grad()
        g1 = torch.randn(p1.shape, generator=g)
        g2 = torch.randn(p2.shape, generator=g) * (10.0 ** (step - 3))
        p1.grad, p2.grad = g1.clone(), g2.clone()
        so.step_and_update_lr()
        lrs.append(so._optimizer.param_groups[0]["lr"])
        arrs.update({f"g1_{step}": npy(g1), f"g2_{step}": npy(g2), f"p1_{step}": npy(p1), f"p2_{step}": npy(p2)})
    arrs["lrs"] = np.array(lrs, dtype=np.float64)
    save("optim", **arrs)


def gen_train3():
    torch.manual_seed(51)
    g = torch.Generator().manual_seed(52)
    vocab = (29, 31)
    m = Transformer(vocab[0], vocab[1], tgt_emb_prj_weight_sharing=False, emb_src_tgt_weight_sharing=False, **TINY)
    xavier_like_initiate(m)
    randomize_small_params(m, g)
    # dropout disabled the way freeze flags would (eval mode): device RNG streams cannot be matched
    m.eval()
    so = ScheduledOptim(optim.Adam(filter(lambda x: x.requires_grad, m.parameters()), betas=(0.9, 0.98), eps=1e-09),
                        TINY["d_model"], 3)
    arrs = {"vocab": np.array(vocab, dtype=np.int64), "n_warmup": np.int64(3)}
    arrs.update(sd_arrays("w0:", m))
    losses, ncs, nws = [], [], []
    for step in range(4):
        src_seq, src_pos = make_batch(g, 5, 2, 8, vocab[0])
        tgt_seq, tgt_pos = make_batch(g, 5, 2, 8, vocab[1])
        so.zero_grad()
        pred = m(src_seq, src_pos, tgt_seq, tgt_pos)
        gold = tgt_seq[:, 1:]
        loss, nc = performance(pred.view(-1, pred.size(2)), gold, True)
        loss.backward()
        so.step_and_update_lr()
        losses.append(loss.item()); ncs.append(nc); nws.append(gold.ne(PAD).sum().item())
        arrs.update({f"src_seq_{step}": npy(src_seq), f"src_pos_{step}": npy(src_pos),
                     f"tgt_seq_{step}": npy(tgt_seq), f"tgt_pos_{step}": npy(tgt_pos)})
    arrs.update(sd_arrays("w4:", m))
    arrs["losses"] = np.array(losses, dtype=np.float64)
    arrs["ncorrect"] = np.array(ncs, dtype=np.int64)
    arrs["nwords"] = np.array(nws, dtype=np.int64)
    save("train4", **arrs)


def gen_deep_train30():
    """A deep (6 + 6 layer) post-LN stack through 30 optimiser steps of a copy task (target = source), the
    configuration the trainer is benchmarked in scaled down to d = 32: per-step loss, accuracy and token counts
    from the reference's own modules, initial weights and every batch.  Eval-mode dropout as in train4."""
    torch.manual_seed(61)
    g = torch.Generator().manual_seed(62)
    vocab = 40
    cfg = dict(TINY)
    cfg.update(n_layers=6, len_max_seq=14)
    m = Transformer(vocab, vocab, tgt_emb_prj_weight_sharing=False, emb_src_tgt_weight_sharing=False, **cfg)
    xavier_like_initiate(m)
    m.eval()
    n_warm = 10
    so = ScheduledOptim(optim.Adam(filter(lambda x: x.requires_grad, m.parameters()), betas=(0.9, 0.98), eps=1e-09),
                        cfg["d_model"], n_warm)
    arrs = {"vocab": np.array([vocab, vocab], dtype=np.int64), "n_warmup": np.int64(n_warm), "n_layers": np.int64(6),
            "len_max_seq": np.int64(14)}
    arrs.update(sd_arrays("w0:", m))
    losses, ncs, nws = [], [], []
    for step in range(30):
        seq, pos = make_batch(g, 16, 3, 10, vocab)
        so.zero_grad()
        pred = m(seq, pos, seq, pos)
        gold = seq[:, 1:]
        loss, nc = performance(pred.view(-1, pred.size(2)), gold, True)
        loss.backward()
        so.step_and_update_lr()
        losses.append(loss.item()); ncs.append(nc); nws.append(gold.ne(PAD).sum().item())
        arrs.update({f"seq_{step}": npy(seq), f"pos_{step}": npy(pos)})
    arrs["losses"] = np.array(losses, dtype=np.float64)
    arrs["ncorrect"] = np.array(ncs, dtype=np.int64)
    arrs["nwords"] = np.array(nws, dtype=np.int64)
    save("deep_train30", **arrs)


def gen_beam():
    """Copy-task model with healthy top-k margins, then the reference beam search on it."""
    torch.manual_seed(61)
    g = torch.Generator().manual_seed(62)
    V = 24
    cfg = dict(TINY)
    cfg["len_max_seq"] = 16
    m = Transformer(V, V, tgt_emb_prj_weight_sharing=False, emb_src_tgt_weight_sharing=False, **cfg)
    xavier_like_initiate(m)
    m.train()
    opt = optim.Adam(filter(lambda x: x.requires_grad, m.parameters()), lr=2e-3, betas=(0.9, 0.98), eps=1e-9)
    for it in range(900):
        src_seq, src_pos = make_batch(g, 32, 1, 9, V)
        opt.zero_grad()
        pred = m(src_seq, src_pos, src_seq, src_pos)
        loss = calculate_loss(pred.view(-1, V), src_seq[:, 1:], True)
        (loss / src_seq[:, 1:].ne(PAD).sum()).backward()
        opt.step()
        if it % 300 == 0:
            print("  copy-task step", it, "loss/word", loss.item() / src_seq[:, 1:].ne(PAD).sum().item())
    m.eval()
    src_seq, src_pos = make_batch(g, 6, 1, 9, V)
    arrs = {"src_seq": npy(src_seq), "src_pos": npy(src_pos), "vocab": np.int64(V),
            "len_max_seq": np.int64(cfg["len_max_seq"])}
    arrs.update(sd_arrays("w:", m))
    for bm, nb in ((1, 1), (3, 2), (5, 1), (5, 3)):
        opt_ns = Namespace(cuda=False, cuda_device=None, beam_size=bm, n_best=nb)
        tr = RefTranslatorMod.Translator(opt_ns, new=False)
        tr.model = m
        tr.max_token_seq_len = cfg["len_max_seq"]
        hyps, scores = tr.translate_batch(src_seq, src_pos)
        maxlen = max(len(h) for hs in hyps for h in hs)
        H = np.full((len(hyps), nb, maxlen), -1, dtype=np.int64)
        Ln = np.zeros((len(hyps), nb), dtype=np.int64)
        S = np.zeros((len(hyps), nb), dtype=np.float32)
        for i, hs in enumerate(hyps):
            for j, h in enumerate(hs):
                H[i, j, : len(h)] = h
                Ln[i, j] = len(h)
                S[i, j] = float(scores[i][j])
        arrs[f"hyp_b{bm}_n{nb}"] = H
        arrs[f"len_b{bm}_n{nb}"] = Ln
        arrs[f"score_b{bm}_n{nb}"] = S
        print(f"  beam {bm} n_best {nb}: lens {Ln[:, 0].tolist()} src lens {(src_seq != 0).sum(1).tolist()}")
    # per-step log-probs of the first decode step, to pin "generator without x_logit_scale"
    with torch.no_grad():
        enc, = m.encoder(src_seq, src_pos)
        dec_seq = torch.full((src_seq.size(0), 1), SOS, dtype=torch.long)
        dec_pos = torch.ones_like(dec_seq)
        dec, = m.decoder(dec_seq, dec_pos, src_seq, enc)
        arrs["step1_logp"] = npy(F.log_softmax(m.generator(dec[:, -1, :]), dim=1))
    save("beam", **arrs)


def det_weights(shapes, seed):
    """Deterministic weights any side can rebuild from (name, shape, seed) alone."""
    out = {}
    for i, (k, shp) in enumerate(shapes):
        rng = np.random.default_rng(seed + i)
        if len(shp) == 1:
            base = 1.0 if k.endswith("layer_norm.weight") else 0.0
            out[k] = (base + 0.05 * rng.standard_normal(shp)).astype(np.float32)
        else:
            fan = shp[0] + shp[1]
            out[k] = (rng.standard_normal(shp) * np.sqrt(2.0 / fan)).astype(np.float32)
    return out


def gen_cfg1_checksum():
    """cfg-1 sized (2-layer d=256 h=8 F=1024 V=4096, B=64) forward + loss; weights are rebuilt
    from a seed on the test side, only checksums and a logits slice are stored."""
    V, Lmax = 4096, 52
    m = Transformer(V, V, Lmax, d_word_vec=256, d_model=256, d_inner=1024, n_layers=2, n_head=8, d_k=32, d_v=32,
                    tgt_emb_prj_weight_sharing=False, emb_src_tgt_weight_sharing=False)
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items() if "position_enc" not in k]
    w = det_weights(shapes, 1234)
    sd = m.state_dict()
    for k, v in w.items():
        sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(7)
    src_seq, src_pos = make_batch(g, 64, 6, 24, V)
    tgt_seq, tgt_pos = make_batch(g, 64, 6, 24, V)
    # ReLU kink margins: the smallest |pre-activation| on non-pad rows of every FFN, in forward
    # order [enc0, enc1, dec0, dec1].  An fp32 implementation with another summation order can
    # land on the other side of a kink that is ~1e-7 away, which changes the gradient of that
    # element by O(1) -- the parity test uses these margins to know which gradients are exposed.
    margins = {}
    ffns = [("enc%d" % i, l.pos_ffn, src_seq) for i, l in enumerate(m.encoder.layer_stack)] + \
           [("dec%d" % i, l.pos_ffn, tgt_seq[:, :-1]) for i, l in enumerate(m.decoder.layer_stack)]
    hooks = []
    for name, ffn, seq in ffns:
        def hook(mod, inp, out, name=name, seq=seq):
            pre = out.detach().transpose(1, 2)[seq.ne(PAD)]
            margins[name] = pre.abs().min().item()
        hooks.append(ffn.w_1.register_forward_hook(hook))
    logits = m(src_seq, src_pos, tgt_seq, tgt_pos)
    for h in hooks:
        h.remove()
    gold = tgt_seq[:, 1:]
    loss, nc = performance(logits.view(-1, V), gold, True)
    loss.backward()
    lg = npy(logits)
    grad_slices = {}
    for k, p in m.named_parameters():
        if p.grad is not None and (k.endswith("slf_attn.w_qs.weight") or k.endswith("pos_ffn.w_1.weight")
                                   or k.endswith("pos_ffn.w_2.bias") or k.endswith("enc_attn.w_vs.weight")):
            grad_slices["gs:" + k] = npy(p.grad)[:8].copy()
    print("  ReLU kink margins:", margins)
    save("cfg1_checksum",
         src_seq=npy(src_seq), src_pos=npy(src_pos), tgt_seq=npy(tgt_seq), tgt_pos=npy(tgt_pos),
         shapes_names=np.array([k for k, _ in shapes]), seed=np.int64(1234),
         logits_sum=np.float64(lg.astype(np.float64).sum()), logits_abs_sum=np.float64(np.abs(lg.astype(np.float64)).sum()),
         logits_slice=lg[:, :, :8].copy(), loss=npy(loss), ncorrect=np.int64(nc),
         g_gen_slice=npy(m.generator.weight.grad)[:16].copy(),
         g_src_emb_norm=np.float64(m.encoder.src_word_emb.weight.grad.double().norm().item()),
         relu_margins=np.array([margins[n] for n in ("enc0", "enc1", "dec0", "dec1")], dtype=np.float64),
         **grad_slices)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    if len(sys.argv) > 1:  # regenerate selected fixtures only, e.g. `make_golden.py cfg1_checksum`
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_attention()
    gen_sublayers()
    gen_sinusoid()
    gen_transformer()
    gen_loss()
    gen_optim()
    gen_train3()
    gen_deep_train30()
    gen_beam()
    gen_cfg1_checksum()
