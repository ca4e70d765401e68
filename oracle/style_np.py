"""CPU restatement (numpy) of the reference's style-transfer decoder objective — TEST INFRASTRUCTURE ONLY.

Nothing in the product path imports this file; tests use it as the checker for the HIP path.
Parity pinned by tests/golden/style_{trim,pad}.npz, which tools/make_golden_style.py wrote by running
the reference's Transformer and ConvNet classes.

What it restates:
  ConvNet.forward                     base/classifier/onmt/CNNModels.py:48-65
  classifier_loss                     base/train_decoder.py:385-424
  tf_forward / performance / one training step of compute_epoch
                                      base/train_decoder.py:360-370, 427-444, 447-497
"""
import numpy as np

from . import transformer_np as T

PAD = 0


def softmax_rows(x):
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


def convnet_fwd(cnn, x, filter_size):
    """x (seq, B, V) class probabilities -> (B,) sigmoid outputs.  CNNModels.py:48-65: the embedding lookup
    is a product with the table, the convolution spans `filter_size` steps of all embedding channels, the
    pooling window (sequence_length - filter_size + 1, set at construction, :18,36-38) covers every
    position of an input of sequence_length steps; dropout is inactive (eval mode, train_decoder.py:350)."""
    S, B, V = x.shape
    emb = (x.reshape(-1, V) @ cnn["word_lut.weight"]).reshape(S, B, -1)  # (S, B, E)
    w = cnn["conv1.weight"][..., 0]  # (F, E, fs)
    P = S - filter_size + 1
    h = np.zeros((B, w.shape[0], P), dtype=x.dtype)
    for t in range(P):
        win = emb[t:t + filter_size]  # (fs, B, E)
        h[:, :, t] = np.einsum("jbe,fej->bf", win, w) + cnn["conv1.bias"]
    h = np.maximum(h, 0).max(-1)  # ReLU, max over time
    lin = h @ cnn["linear.weight"].T + cnn["linear.bias"]  # (B, 1)
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-lin[:, 0]))


def classifier_loss(dec_output, class_w, class_b, cnn, seq_length, filter_size, target_label):
    """train_decoder.py:385-424: Linear -> softmax over the classifier's vocabulary -> time-major ->
    zero-pad or trim to the classifier's window -> ConvNet -> mean binary cross entropy against the target
    label (nn.BCELoss clamps each logarithm at -100).  Returns (loss, class_outputs (B,))."""
    B, L, d = dec_output.shape
    out = softmax_rows(dec_output @ class_w.T + class_b).transpose(1, 0, 2)  # (L, B, V)
    if L < seq_length:
        out = np.concatenate([out, np.zeros((seq_length - L, B, out.shape[2]), dtype=out.dtype)], 0)
    else:
        out = out[:seq_length]
    y = convnet_fwd(cnn, out, filter_size)
    with np.errstate(divide="ignore"):
        log_y, log_1my = np.maximum(np.log(y), -100.0), np.maximum(np.log(1.0 - y), -100.0)
    loss = -(target_label * log_y + (1 - target_label) * log_1my).mean()
    return loss, y


def style_step(w, cfg, batch, class_w, class_b, cnn, seq_length, filter_size, target_label, smoothing):
    """One training-branch iteration of compute_epoch (train_decoder.py:462-477) up to the gradients.
    The classifier sees `pred.data` (:399), so no gradient reaches the decoder through it:
    d(net_loss)/d(parameters) = l_class * d(l_recon)/d(parameters); the encoder output is detached (:368),
    so encoder parameters get none (a shared embedding still gets the decoder's and generator's share).
    Returns (net_loss, l_recon, l_class, n_correct, class_outputs, grads, dec_output)."""
    src_seq, src_pos, tgt_seq, tgt_pos = batch
    gold = tgt_seq[:, 1:]
    tgt_in, tgt_pos_in = tgt_seq[:, :-1], tgt_pos[:, :-1]
    enc_out, _, _ = T.encoder_fwd(w, cfg, src_seq, src_pos)
    dec_out, _, _, dec_caches = T.decoder_fwd(w, cfg, tgt_in, tgt_pos_in, src_seq, enc_out)
    l_class, y = classifier_loss(dec_out, class_w, class_b, cnn, seq_length, filter_size, target_label)
    logits = (dec_out @ w["generator.weight"].T) * cfg.x_logit_scale
    V = logits.shape[-1]
    l_recon, dlogits, n_correct = T.ls_xent(logits.reshape(-1, V), gold.reshape(-1), smoothing)
    dl = dlogits.reshape(logits.shape) * l_class * cfg.x_logit_scale
    grads = {}
    T._acc(grads, "generator.weight", dl.reshape(-1, V).T @ dec_out.reshape(-1, dec_out.shape[-1]))
    T.decoder_bwd(w, cfg, tgt_in, dl @ w["generator.weight"], dec_caches, grads)
    return l_recon * l_class, l_recon, l_class, n_correct, y, grads, dec_out
