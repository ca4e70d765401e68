#!/usr/bin/env python3
"""Golden fixtures for the style-transfer decoder objective (base/train_decoder.py), produced by
running the reference's own network and classifier classes.

Runs ONLY in the build container.  Imported unmodified from the reference:
  lib.transformer.Models.Transformer, lib.transformer.Optim.ScheduledOptim,
  classifier/onmt/CNNModels.ConvNet
train_decoder.py itself cannot be imported (it imports lib.TransformerModel, which needs the absent
packages telebot / toolz / nltk, and it runs argument parsing at import); the few functions of its
training step are restated from the source text, line for line, with the three changes modern torch
forces:
  * `Variable(pred.data, requires_grad=True, volatile=False)` (train_decoder.py:399) is
    `pred.detach().requires_grad_(True)` — the same leaf cut off from the decoder;
  * `.cuda()` / `torch.cuda.FloatTensor` are the CPU equivalents;
  * `criterion2` is passed in instead of being a module global.
Known defect worked around: when the target is longer than the classifier's window the reference slices a
transposed tensor and `ConvNet.forward` calls `.view` on it (CNNModels.py:52), which raises; the slice is made
contiguous here, nothing else changes.
The fixtures are plain .npz data.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_style.py
"""
import os
import sys
from argparse import Namespace

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/base")
sys.path.insert(0, "/root/reference/base/classifier")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import torch.optim as optim  # noqa: E402

import onmt.CNNModels as CNNModels  # noqa: E402
from lib.transformer.Models import Transformer  # noqa: E402
from lib.transformer.Optim import ScheduledOptim  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
PAD, SOS, EOS = 0, 2, 3
TINY = dict(len_max_seq=12, d_word_vec=32, d_model=32, d_inner=64, n_layers=2, n_head=4, d_k=8, d_v=8, dropout=0.0)


def npy(t):
    return t.detach().cpu().numpy().copy()


def make_batch(g, n, lo, hi, vocab):
    """[SOS] ids [EOS] right-padded with PAD; pos = 1..len, 0 on pads (base/core/dataset.py:21-46)."""
    lens = torch.randint(lo, hi + 1, (n,), generator=g).tolist()
    L = max(lens) + 2
    seq = torch.zeros(n, L, dtype=torch.long)
    pos = torch.zeros(n, L, dtype=torch.long)
    for i, l in enumerate(lens):
        seq[i, 0] = SOS
        seq[i, 1:l + 1] = torch.randint(5, vocab, (l,), generator=g)
        seq[i, l + 1] = EOS
        pos[i, :l + 2] = torch.arange(1, l + 3)
    return seq, pos


def calculate_loss(pred, gold, smoothing):
    # base/lib/TransformerModel.py:353-375
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


def tf_forward(model, src_seq, src_pos, tgt_seq, tgt_pos):
    # base/train_decoder.py:360-370
    tgt_seq, tgt_pos = tgt_seq[:, :-1], tgt_pos[:, :-1]
    enc_output, _ = model.encoder(src_seq, src_pos, True)
    dec_out = model.decoder(tgt_seq, tgt_pos, src_seq, enc_output.detach(), True)
    dec_output, _, _ = dec_out
    return dec_output


def n_tokens_correct(pred, gold, pad_token):
    # base/train_decoder.py:373-382
    pred = pred.max(1)[1]
    gold = gold.contiguous().view(-1)
    non_pad_mask = gold.ne(pad_token)
    n_correct = pred.eq(gold)
    return n_correct.masked_select(non_pad_mask).sum().item()


def classifier_loss(pred, class_model, class_input, criterion2):
    # base/train_decoder.py:385-424
    softmax = nn.Softmax(dim=1)
    seq_length = class_model.opt.sequence_length
    target_label = class_model.refs.tgt_label
    exp = pred.detach().requires_grad_(True)
    linear = class_input(exp)
    d0, d1, d2 = linear.size()
    out = softmax(linear.view(-1, d2)).view(d0, d1, d2)
    out = out.transpose(0, 1)
    d0, d1, d2 = out.size()
    if d0 < seq_length:
        cnn_padding = torch.zeros(abs(seq_length - d0), d1, d2)
        cat = torch.cat((out, cnn_padding), 0)
    else:
        # defect worked around: the reference hands this non-contiguous slice to `.view`
        # (classifier/onmt/CNNModels.py:52), which raises; made contiguous, like the other branch's result
        cat = out[:seq_length].contiguous()
    class_outputs = class_model(cat).squeeze()
    return criterion2(class_outputs, torch.full((d1,), float(target_label))), class_outputs


def performance(model, pred_before, gold, class_input, class_model, criterion2, smoothing):
    # base/train_decoder.py:427-444
    pred = model.generator(pred_before) * model.x_logit_scale
    l_classifier, class_outputs = classifier_loss(pred_before, class_model, class_input, criterion2)
    pred = pred.view(-1, pred.size(2))
    l_reconstruction = calculate_loss(pred, gold, smoothing)
    n_correct = n_tokens_correct(pred, gold, pad_token=PAD)
    return l_reconstruction, l_classifier, n_correct, class_outputs


def gen_case(tag, seed, seq_length, target_label, smoothing, tied):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    vocab = (31, 31) if tied else (29, 31)
    model = Transformer(vocab[0], vocab[1], tgt_emb_prj_weight_sharing=tied, emb_src_tgt_weight_sharing=tied, **TINY)
    for p in model.parameters():  # base/lib/TransformerModel.py:73-75
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    # base/train_decoder.py:332-353: the classifier is built from its saved options; its vocabulary is
    # the translation model's
    class_opt = Namespace(num_filters=6, filter_size=3, sequence_length=seq_length, word_vec_size=8, dropout=0.2,
                          num_classes=2)
    class_model = CNNModels.ConvNet(class_opt, vocab[1])
    class_model.eval()
    class_model.refs = Namespace(tgt_label=target_label)
    with torch.no_grad():  # spread the classifier's outputs away from 0.5
        class_model.word_lut.weight.mul_(1.5)
        class_model.linear.weight.mul_(4.0)
    model_dim = model.decoder.layer_stack[0].slf_attn.w_qs.out_features  # train_decoder.py:546
    class_input = nn.Sequential(nn.Linear(model_dim, class_model.word_lut.weight.shape[0]))  # :549-551
    with torch.no_grad():
        class_input[0].weight.mul_(6.0)
    criterion2 = nn.BCELoss()
    # base/lib/TransformerModel.py:155-168, called before the classifier exists (train_decoder.py:536)
    so = ScheduledOptim(optim.Adam(filter(lambda x: x.requires_grad, model.parameters()), betas=(0.9, 0.98), eps=1e-09),
                        TINY["d_model"], 3)
    model.encoder.eval()  # train_decoder.py:539

    arrs = {"vocab": np.array(vocab, dtype=np.int64), "n_warmup": np.int64(3), "seq_length": np.int64(seq_length),
            "target_label": np.int64(target_label), "smoothing": np.int64(smoothing), "tied": np.int64(tied),
            "num_filters": np.int64(6), "filter_size": np.int64(3), "word_vec_size": np.int64(8)}
    arrs.update({"w0:" + k: npy(v) for k, v in model.state_dict().items()})
    arrs.update({"cnn:" + k: npy(v) for k, v in class_model.state_dict().items()})
    arrs["class_input.weight"] = npy(class_input[0].weight)
    arrs["class_input.bias"] = npy(class_input[0].bias)
    for step in range(3):
        src_seq, src_pos = make_batch(g, 4, 2, 7, vocab[0])
        tgt_seq, tgt_pos = make_batch(g, 4, 2, 7, vocab[1])
        # base/train_decoder.py:447-497, training branch
        # the reference puts decoder and generator in train mode here; the fixture keeps them in eval mode so that
        # no dropout mask is drawn (random streams cannot be matched across devices) — the attention
        # dropout of 0.1 is a constructor default that no option turns off
        model.decoder.eval()
        model.generator.eval()
        model.zero_grad()
        gold = tgt_seq[:, 1:]
        pred = tf_forward(model, src_seq, src_pos, tgt_seq, tgt_pos)
        l_recon, l_class, n_correct, class_outputs = performance(model, pred, gold, class_input, class_model,
                                                                 criterion2, smoothing)
        net_loss = l_recon * l_class
        net_loss.backward()
        if step == 0:
            arrs["dec_output_0"] = npy(pred)
            for k, p in model.named_parameters():
                if p.grad is not None:
                    arrs["g0:" + k] = npy(p.grad)
            arrs["has_encoder_grad"] = np.int64(any(p.grad is not None for p in model.encoder.layer_stack.parameters()))
        so.step_and_update_lr()
        arrs.update({f"src_seq_{step}": npy(src_seq), f"src_pos_{step}": npy(src_pos), f"tgt_seq_{step}": npy(tgt_seq),
                     f"tgt_pos_{step}": npy(tgt_pos), f"class_outputs_{step}": npy(class_outputs.reshape(-1))})
        arrs[f"losses_{step}"] = np.array([net_loss.item(), l_recon.item(), l_class.item(), n_correct], dtype=np.float64)
    arrs.update({"w3:" + k: npy(v) for k, v in model.state_dict().items()})
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{tag}.npz  {os.path.getsize(path) / 1024:.1f} KiB  losses {[arrs[f'losses_{s}'].round(4).tolist() for s in range(3)]}")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    # target length after dropping the last column is <= 8: one classifier window shorter (trim), one longer (pad)
    gen_case("style_trim", 71, seq_length=5, target_label=1, smoothing=True, tied=False)
    gen_case("style_pad", 73, seq_length=11, target_label=0, smoothing=False, tied=True)
