"""What the one device -> host read per training step (row counts of the packing plan) costs: the benchmark step
with the plan computed in the step (as shipped) against the same step with the plans of the four resident batches
looked up from a table (no host read, the host runs ahead of the GPU).  Interleaved rounds in one process."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import ops
from specialk_amd.lib.transformer import Models

cfg = bench.CONFIGS[os.environ.get("SK_AB_CONFIG", "cfg2")]
dev = torch.device("cuda", 0)
m, opt = bench.build(cfg, torch.bfloat16, dev)
ops.manual_seed(1000)
g = torch.Generator().manual_seed(7)
batches = []
for _ in range(4):
    s, sp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
    t, tp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
    batches.append(tuple(x.to(dev) for x in (s, sp, t, tp)))

orig = Models.plan_packing
table = {}


def cached(seqs, trims, check=None):
    key = tuple(s.data_ptr() for s in seqs)
    if key not in table:
        table[key] = orig(seqs, trims, check)
    return table[key]


def run(fn, steps=8):
    Models.plan_packing = fn
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        bench.train_step(m, opt, batches[i % 4], None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for fn in (orig, cached):
    run(fn, 4)
res = {"in-step plan": [], "table": []}
for rnd in range(6):
    res["in-step plan"].append(run(orig))
    res["table"].append(run(cached))
for k, v in res.items():
    v.sort()
    print(f"{k:14s} median {v[len(v) // 2]:7.3f} ms/step  min {v[0]:7.3f}  max {v[-1]:7.3f}")
