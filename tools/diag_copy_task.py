"""Does the trainer learn?  A small Transformer (2+2 layers, d=256) on a synthetic copy task, in fp32
(the parity path) and in bf16 (the fast path, grouped weight gradients), same data and seeds:
loss and token accuracy every 100 steps.  Usage: diag_copy_task.py [steps] [warmup] [cfg1|cfg2]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg = dict(bench.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "cfg1"]); cfg["hi"] = 30; cfg["tied"] = bool(os.environ.get("SK_DIAG_TIED")) or cfg["tied"]; cfg["n_layers"] = int(os.environ.get("SK_DIAG_LAYERS", cfg["n_layers"]))
for dtype in ((torch.bfloat16,) if os.environ.get("SK_DIAG_BF16_ONLY") else (torch.float32, torch.bfloat16)):
    m, opt = bench.build(cfg, dtype, "cuda")
    opt.n_warmup_steps = warmup
    opt.init_lr *= float(os.environ.get("SK_DIAG_LR_SCALE", "1"))
    ops.manual_seed(5)
    g = torch.Generator().manual_seed(3)
    t0 = time.perf_counter()
    for it in range(steps):
        s_, sp_ = bench.make_batch(g, 256, 6, 24, 500)
        batch = tuple(x.cuda() for x in (s_, sp_, s_, sp_))
        stats = bench.train_step(m, opt, batch, None)
        if (it + 1) % 100 == 0:
            st = stats.tolist()
            print(f"{str(dtype):15s} step {it + 1}: loss/token {st[0] / st[2]:.3f}  accuracy {st[1] / st[2]:.3f}  ({time.perf_counter() - t0:.1f} s)", flush=True)
