"""Diagnostic (GPU box): per-tensor error of the fp32 HIP path against the fp64 numpy oracle on
the cfg-1 sized model.  Prints max-abs error relative to each tensor's max."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from oracle import transformer_np as O
from specialk_amd import ops
from specialk_amd.lib.transformer.Models import Transformer

g = load_golden("cfg1_checksum")
V = 4096
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
w = {k: v.numpy().astype(np.float64) for k, v in m.state_dict().items()}
cfg = O.Config(2, 8, 32, 32, 256, 1024)
b = [g[k] for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")]
logits, cache = O.transformer_fwd(w, cfg, *b)
loss, dpred, nc = O.ls_xent(logits.reshape(-1, V), b[2][:, 1:].reshape(-1), True)
grads = O.transformer_bwd(w, cfg, dpred.reshape(logits.shape), cache)

m = m.cuda().eval()
bd = [torch.from_numpy(x).cuda() for x in b]
rel = lambda a, r: np.abs(a - r).max() / np.abs(r).max()
rl2 = lambda a, r: np.linalg.norm((a - r).ravel()) / np.linalg.norm(r.ravel())
for dt in (torch.float32, torch.bfloat16):
    m.set_compute_dtype(dt)
    m.zero_grad()
    lg = m(*bd)
    l, stats = ops.xent(lg.view(-1, V), bd[2][:, 1:], True)
    l.backward()
    print("====", dt, "logits max/L2", rel(lg.detach().cpu().double().numpy(), logits), rl2(lg.detach().cpu().double().numpy(), logits),
          "loss", abs(l.item() - loss) / loss)
    for k, p in m.named_parameters():
        if p.grad is not None and k in grads and not k.endswith("w_ks.bias"):
            a, r = p.grad.cpu().double().numpy(), grads[k].reshape(p.shape)
            print(f"{k:55s} max {rel(a, r):.3e}  L2 {rl2(a, r):.3e}")
