"""The numpy restatement of the style-transfer decoder objective against the golden vectors produced by
the reference's Transformer and ConvNet classes (CPU only)."""
import numpy as np
import pytest

from conftest import load_golden, split_prefixed
from oracle import style_np as S
from oracle import transformer_np as O
from test_oracle import close, tiny_cfg


def style_inputs(g):
    cnn = split_prefixed(g, "cnn:")
    kw = dict(class_w=g["class_input.weight"], class_b=g["class_input.bias"], cnn=cnn,
              seq_length=int(g["seq_length"]), filter_size=int(g["filter_size"]),
              target_label=int(g["target_label"]), smoothing=bool(g["smoothing"]))
    return kw


@pytest.mark.parametrize("name", ["style_trim", "style_pad"])
def test_style_steps(name):
    g = load_golden(name)
    tied = bool(g["tied"])
    w = {k: v.copy() for k, v in split_prefixed(g, "w0:").items()}
    cfg = tiny_cfg(tied)
    kw = style_inputs(g)
    trainable = [k for k in w if "position_enc" not in k]
    m = {k: np.zeros_like(w[k]) for k in trainable}
    v = {k: np.zeros_like(w[k]) for k in trainable}
    assert int(g["has_encoder_grad"]) == 0
    for step in range(3):
        if tied:  # one matrix under three names (lib/transformer/Models.py:259-271)
            w["generator.weight"] = w["encoder.src_word_emb.weight"] = w["decoder.tgt_word_emb.weight"]
        b = [g[f"{n}_{step}"] for n in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")]
        net, l_recon, l_class, nc, y, grads, dec_out = S.style_step(w, cfg, b, **kw)
        ref = g[f"losses_{step}"]
        assert abs(net - ref[0]) <= 3e-5 * abs(ref[0]) and abs(l_recon - ref[1]) <= 3e-5 * abs(ref[1])
        assert abs(l_class - ref[2]) <= 3e-5 * abs(ref[2]) and nc == int(ref[3])
        close(y, g[f"class_outputs_{step}"])
        grads = O.tie_grads({**{k: np.zeros_like(w[k]) for k in ("encoder.src_word_emb.weight",)}, **grads}, tied, tied)
        if step == 0:
            close(dec_out, g["dec_output_0"])
            g0 = split_prefixed(g, "g0:")
            for k, ref_g in g0.items():
                if "position_enc" in k:
                    continue
                close(grads[k], ref_g, rtol=5e-5, atol=1e-5)
            # the detached encoder output leaves the encoder's own layers without gradients
            assert not any(k.startswith("encoder.layer_stack") for k in g0)
        lr = np.float32(O.noam_lr(step + 1, cfg.d_model, int(g["n_warmup"])))
        done = set()
        for k in trainable:
            if k not in grads or (not tied and k.startswith("encoder.")) or id(w[k]) in done:
                continue
            done.add(id(w[k]))
            O.adam_step(w[k], grads[k], m[k], v[k], step + 1, lr)
    w3 = split_prefixed(g, "w3:")
    for k in trainable:
        if k.endswith("w_ks.bias"):  # true gradient 0: Adam turns rounding noise into +-lr steps
            continue
        close(w[k], w3[k], rtol=3e-4)


def test_classifier_window_edges():
    """Windows shorter and longer than the sentence, and a saturated sigmoid (BCELoss clamps log at -100)."""
    rng = np.random.default_rng(3)
    B, L, d, V, E, F, fs = 3, 6, 8, 11, 4, 5, 3
    cnn = {"word_lut.weight": rng.standard_normal((V, E)), "conv1.weight": rng.standard_normal((F, E, fs, 1)),
           "conv1.bias": rng.standard_normal(F), "linear.weight": rng.standard_normal((1, F)), "linear.bias": np.zeros(1)}
    x = rng.standard_normal((B, L, d))
    cw, cb = rng.standard_normal((V, d)), rng.standard_normal(V)
    for seq_length in (3, 6, 9):
        loss, y = S.classifier_loss(x, cw, cb, cnn, seq_length, fs, 1)
        assert y.shape == (B,) and np.isfinite(loss)
    cnn["linear.bias"] = np.array([-1e4])
    loss, y = S.classifier_loss(x, cw, cb, cnn, 6, fs, 1)
    assert loss == 100.0
