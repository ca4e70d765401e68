"""Prints, for the cfg-1 sized fixture in bf16, each checked gradient slice's max-abs and L2
error relative to the fp32 reference values, once per GEMM kernel choice (SKHIP_GEMM_STAGED)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, split_prefixed  # noqa: E402


def run(tag):
    from specialk_amd import ops
    from specialk_amd.lib.transformer.Models import Transformer
    g = load_golden("cfg1_checksum")
    V = 4096
    m = Transformer(V, V, 52, d_word_vec=256, d_model=256, d_inner=1024, n_layers=2, n_head=8, d_k=32, d_v=32,
                    tgt_emb_prj_weight_sharing=False, emb_src_tgt_weight_sharing=False)
    sd = m.state_dict()
    seed = int(g["seed"])
    for i, k in enumerate(g["shapes_names"].tolist()):
        shp = tuple(sd[k].shape)
        rng = np.random.default_rng(seed + i)
        if len(shp) == 1:
            base = 1.0 if k.endswith("layer_norm.weight") else 0.0
            sd[k] = torch.from_numpy((base + 0.05 * rng.standard_normal(shp)).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((rng.standard_normal(shp) * np.sqrt(2.0 / (shp[0] + shp[1]))).astype(np.float32))
    m.load_state_dict(sd)
    m = m.cuda().eval().set_compute_dtype(torch.bfloat16)
    b = [torch.from_numpy(g[k]).cuda() for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")]
    logits = m(*b)
    loss, _ = ops.xent(logits.view(-1, V), b[2][:, 1:], True)
    loss.backward()
    named = dict(m.named_parameters())
    print(f"--- {tag}: loss rel err {abs(loss.item() - float(g['loss'])) / float(g['loss']):.2e}")
    for k, ref in split_prefixed(g, "gs:").items():
        got = named[k].grad[:8].double().cpu().numpy()
        ref = ref.astype(np.float64)
        print(f"{k:50s} max {np.abs(got - ref).max() / np.abs(ref).max():.3f}  l2 {np.linalg.norm(got - ref) / np.linalg.norm(ref):.3f}")


if __name__ == "__main__":
    for v in ("1", "0", ""):
        if v:
            os.environ["SKHIP_GEMM_STAGED"] = v
        else:
            os.environ.pop("SKHIP_GEMM_STAGED", None)
        run("staged=" + (v or "auto"))
