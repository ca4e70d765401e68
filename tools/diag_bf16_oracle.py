"""Per-tensor relative L2 distance between the HIP bf16 gradients and the oracle in bf16-storage mode (and, for
scale, the fp32 golden vectors) on the tiny fixtures and the cfg-1 sized model."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
from conftest import load_golden, split_prefixed
from specialk_amd import ops

def report(m, grads_o, gold=None, top=8):
    rows = []
    for k, p in m.named_parameters():
        if p.grad is None or k not in grads_o or k.endswith("w_ks.bias"):
            continue
        got = p.grad.double().cpu()
        ref = torch.from_numpy(np.asarray(grads_o[k], dtype=np.float64)).reshape(p.shape)
        r = ((got - ref).norm() / ref.norm()).item()
        r32 = None
        if gold is not None and k in gold:
            g32 = torch.from_numpy(np.asarray(gold[k], dtype=np.float64)).reshape(p.shape)
            r32 = ((got - g32).norm() / g32.norm()).item()
        rows.append((r, k, r32))
    rows.sort(reverse=True)
    for r, k, r32 in rows[:top]:
        print(f"   {r:.4f}  {k}" + (f"   (vs fp32 golden {r32:.4f})" if r32 is not None else ""))
    print(f"   median {np.median([r for r, _, _ in rows]):.4f} over {len(rows)} tensors")

for tag in ("transformer_untied", "transformer_tied"):
    g = load_golden(tag); tied = bool(g["tied"])
    m = T.load_sd(T.build_tiny(g, tied), split_prefixed(g, "w:")).eval().set_compute_dtype(torch.bfloat16)
    batch = [T.t(g[k]) for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")]
    for packed in (False, True):
        m.zero_grad()
        if packed:
            logits, gold = m.forward_packed(*batch)
            loss, _ = ops.xent(logits, gold, True)
        else:
            logits = m(*batch)
            loss, _ = ops.xent(logits.view(-1, logits.size(2)), batch[2][:, 1:], True)
        loss.backward()
        lo, loss_o, grads_o = T.oracle_bf16(split_prefixed(g, "w:"), T.tiny_oracle_cfg(tied), [g[k] for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")], True, tied)
        print(tag, "packed" if packed else "padded", "loss", loss.item(), "oracle-bf16", loss_o, "fp32", float(g["loss_ls"]))
        if not packed:
            d = (logits.detach().float().cpu().numpy() - lo); print("   logits max abs diff vs oracle-bf16", np.abs(d).max(), "of", np.abs(lo).max())
        report(m, grads_o, split_prefixed(g, "g_ls:"))

# cfg-1 sized model (same construction as tests/test_model_gpu.py::test_cfg1_checksum)
from specialk_amd.lib.transformer.Models import Transformer
from oracle import transformer_np as O
g = load_golden("cfg1_checksum"); V = 4096
m = Transformer(V, V, 52, d_word_vec=256, d_model=256, d_inner=1024, n_layers=2, n_head=8, d_k=32, d_v=32,
                tgt_emb_prj_weight_sharing=False, emb_src_tgt_weight_sharing=False)
sd = m.state_dict(); seed = int(g["seed"])
for i, k in enumerate(g["shapes_names"].tolist()):
    shp = tuple(sd[k].shape); rng = np.random.default_rng(seed + i)
    if len(shp) == 1:
        sd[k] = torch.from_numpy(((1.0 if k.endswith("layer_norm.weight") else 0.0) + 0.05 * rng.standard_normal(shp)).astype(np.float32))
    else:
        sd[k] = torch.from_numpy((rng.standard_normal(shp) * np.sqrt(2.0 / (shp[0] + shp[1]))).astype(np.float32))
m.load_state_dict(sd)
m = m.to("cuda").eval().set_compute_dtype(torch.bfloat16)
b = [T.t(g[k]) for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")]
w_np = {k: v.numpy() for k, v in sd.items()}
for mode in ("padded", "packed+batched"):
    m.zero_grad(set_to_none=True)
    if mode == "padded":
        logits = m(*b); loss, _ = ops.xent(logits.view(-1, V), b[2][:, 1:], True); loss.backward()
    else:
        logits, gold = m.forward_packed(*b); loss, _ = ops.xent(logits, gold, True, unit_grad=True)
        with ops.batched_weight_grads():
            loss.backward()
    lo, loss_o, grads_o = T.oracle_bf16(w_np, O.Config(2, 8, 32, 32, 256, 1024), [g[k] for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")], True)
    print("cfg1", mode, "loss", loss.item(), "oracle-bf16", loss_o, "fp32", float(g["loss"]))
    report(m, grads_o, None, top=12)
