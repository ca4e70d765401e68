"""How reproducible are gradients under bf16 storage at all?  CPU only.

Runs the numpy oracle's bf16-storage mode twice on the same weights and batch — the second time with a relative
perturbation of `eps` (default 1e-6, i.e. a different fp32 summation order) applied ahead of every bf16 rounding —
and prints the per-tensor relative L2 distance between the two sets of parameter gradients.  That distance is the
floor below which no two bf16 implementations of this model can be expected to agree; the whole-model bf16 bounds
in tests/test_model_gpu.py are set from it.

    python tools/diag_bf16_noise.py [eps]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, split_prefixed  # noqa: E402
from oracle import transformer_np as O  # noqa: E402


def run(w, cfg, batch, tied=False):
    with O.bf16_storage():
        logits, cache = O.transformer_fwd(w, cfg, *batch)
        V = logits.shape[-1]
        loss, dp, _ = O.ls_xent(logits.reshape(-1, V), batch[2][:, 1:].reshape(-1), True)
        return loss, O.tie_grads(O.transformer_bwd(w, cfg, dp.reshape(logits.shape), cache), tied, tied)


def floor(name, w, cfg, batch, eps, tied=False):
    l0, g0 = run(w, cfg, batch, tied)
    exact, rng = O.round_bf16, np.random.default_rng(0)

    def noisy(x):
        x = np.asarray(x, dtype=np.float32)
        return exact(x * (1 + eps * rng.standard_normal(x.shape).astype(np.float32)))
    O.round_bf16 = noisy
    try:
        l1, g1 = run(w, cfg, batch, tied)
    finally:
        O.round_bf16 = exact
    rows = sorted(((float(np.linalg.norm(g1[k].astype(np.float64) - g0[k]) / np.linalg.norm(g0[k])), k)
                   for k in g0 if not k.endswith("w_ks.bias")), reverse=True)
    print(f"{name}: loss {l0:.6f} vs {l1:.6f} (rel {abs(l1 - l0) / abs(l0):.2e})")
    for r, k in rows[:5]:
        print(f"   {r:.4f}  {k}")
    print(f"   median {np.median([r for r, _ in rows]):.4f} over {len(rows)} tensors")


if __name__ == "__main__":
    eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6
    for tag in ("transformer_untied", "transformer_tied"):
        g = load_golden(tag)
        tied = bool(g["tied"])
        cfg = O.Config(2, 4, 8, 8, 32, 64, tied_proj=tied)
        floor(tag, split_prefixed(g, "w:"), cfg, [g[k] for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")], eps, tied)
    import torch
    from specialk_amd.lib.transformer.Models import Transformer
    g = load_golden("cfg1_checksum")
    m = Transformer(4096, 4096, 52, d_word_vec=256, d_model=256, d_inner=1024, n_layers=2, n_head=8, d_k=32, d_v=32,
                    tgt_emb_prj_weight_sharing=False, emb_src_tgt_weight_sharing=False)
    sd, seed = m.state_dict(), int(g["seed"])
    for i, k in enumerate(g["shapes_names"].tolist()):
        shp, rng = tuple(sd[k].shape), np.random.default_rng(seed + i)
        if len(shp) == 1:
            sd[k] = torch.from_numpy(((1.0 if k.endswith("layer_norm.weight") else 0.0)
                                      + 0.05 * rng.standard_normal(shp)).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((rng.standard_normal(shp) * np.sqrt(2.0 / (shp[0] + shp[1]))).astype(np.float32))
    floor("cfg1 (d 256, 2+2 layers, V 4096, B 64)", {k: v.numpy() for k, v in sd.items()}, O.Config(2, 8, 32, 32, 256, 1024),
          [g[k] for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")], eps)
