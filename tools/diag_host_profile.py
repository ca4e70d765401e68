"""Where the host spends a training step: cProfile over a few steps of a bench configuration (the queue is drained
before and after, so GPU time shows only where the host waits for it).  Usage: python tools/diag_host_profile.py [cfg1] [steps]"""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
pending = m.plan_ahead(*batches[0])
for i in range(8):
    nxt = m.plan_ahead(*batches[(i + 1) % 4])
    bench.train_step(m, opt, batches[i % 4], None, plans=pending)
    pending = nxt
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(steps):
    nxt = m.plan_ahead(*batches[(i + 1) % 4])
    bench.train_step(m, opt, batches[i % 4], None, plans=pending)
    pending = nxt
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name}: host {1e3 * (t1 - t0) / steps:.3f} ms per step to queue, {1e3 * (t2 - t0) / steps:.3f} ms per step with the queue drained")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
