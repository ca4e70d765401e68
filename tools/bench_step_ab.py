"""A/B of library switches on the whole benchmark training step, interleaved rounds in ONE process (separate
processes on one box differ by several percent).  Usage:
  python tools/bench_step_ab.py "name1:ENV=1,ENV2=0" "name2:ENV=0" ...   ('-' for an empty environment)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import ops

variants = []
for a in sys.argv[1:]:
    name, _, envs = a.partition(":")
    variants.append((name, dict(kv.split("=") for kv in envs.split(",") if kv and kv != "-")))
keys = sorted({k for _, e in variants for k in e})
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


def set_env(e):
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(e)


for name, e in variants:  # warm-up of every variant (workspaces, code objects)
    set_env(e)
    for i in range(3):
        bench.train_step(m, opt, batches[i % 4], None)
torch.cuda.synchronize()
times = {name: [] for name, _ in variants}
STEPS = 4
for rnd in range(int(os.environ.get("SK_AB_ROUNDS", "6"))):
    for name, e in variants:
        set_env(e)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(STEPS):
            bench.train_step(m, opt, batches[i % 4], None)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / STEPS)
for name, _ in variants:
    t = sorted(times[name])
    print(f"{name:24s} median {t[len(t) // 2]:7.3f} ms/step   min {t[0]:7.3f}   max {t[-1]:7.3f}", flush=True)
