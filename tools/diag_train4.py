"""Prints, for the four-step training fixture, per-parameter error metrics of the final weights
(padded and packed paths): max abs error relative to max|w|, and L2 error relative to the L2 norm
of the four-step update."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, split_prefixed
from test_model_gpu import build_tiny, load_sd, TINY, t
from specialk_amd import ops
from specialk_amd.lib.transformer.Optim import ScheduledOptim
from specialk_amd.optim import FusedAdam

g = load_golden("train4")
for packed in (False, True):
    m = load_sd(build_tiny(g, False), split_prefixed(g, "w0:")).eval()
    so = ScheduledOptim(FusedAdam(filter(lambda x: x.requires_grad, m.parameters()), betas=(0.9, 0.98), eps=1e-09),
                        TINY["d_model"], int(g["n_warmup"]))
    for step in range(4):
        b = [t(g[f"{n}_{step}"]) for n in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")]
        so.zero_grad()
        if packed:
            pred, gold = m.forward_packed(*b); loss, stats = ops.xent(pred, gold, True, unit_grad=True)
        else:
            pred = m(*b); loss, stats = ops.xent(pred.view(-1, pred.size(2)), b[2][:, 1:], True, unit_grad=True)
        loss.backward(); so.step_and_update_lr()
    print("packed" if packed else "padded", "n_warmup", int(g["n_warmup"]))
    w0, w4 = split_prefixed(g, "w0:"), split_prefixed(g, "w4:")
    for k, p in m.state_dict().items():
        got = p.double().cpu().numpy(); ref = w4[k].astype(np.float64); upd = ref - w0[k]
        if np.abs(upd).max() == 0: continue
        e = got - ref
        print(f"{k:50s} maxrel {np.abs(e).max()/np.abs(ref).max():.2e}  maxabs {np.abs(e).max():.2e}  upd_max {np.abs(upd).max():.2e}  l2/upd {np.linalg.norm(e)/np.linalg.norm(upd):.2e}  n_bad {(np.abs(e) > 1e-5).sum()}/{e.size}")
