import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from specialk_amd import ops
name = sys.argv[1]
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
m, opt = bench.build(cfg, torch.bfloat16, dev)
ops.manual_seed(1000)
g = torch.Generator().manual_seed(7)
batches = []
for _ in range(4):
    s, sp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
    t, tp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
    batches.append(tuple(x.to(dev) for x in (s, sp, t, tp)))
def run(steps):
    pending = m.plan_ahead(*batches[0])
    for i in range(steps):
        nxt = m.plan_ahead(*batches[(i + 1) % 4])
        bench.train_step(m, opt, batches[i % 4], None, plans=pending)
        pending = nxt
for mode in (True, False, True, False):
    torch.autograd.set_multithreading_enabled(mode)
    run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{name} multithreading={mode}: {1e3 * (t1 - t0) / 60:.3f} ms/step")
