"""Host time against GPU time of the benchmark training step: how long the host takes to queue a step's launches
(no synchronisation inside the loop, plans one step ahead) and how long the GPU takes to run them."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import ops

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
pending = m.plan_ahead(*batches[0])
for i in range(5):
    nxt = m.plan_ahead(*batches[(i + 1) % 4])
    bench.train_step(m, opt, batches[i % 4], None, plans=pending)
    pending = nxt
torch.cuda.synchronize()
N = 20
per_step = []
t0 = time.perf_counter()
for i in range(5, 5 + N):
    h0 = time.perf_counter()
    nxt = m.plan_ahead(*batches[(i + 1) % 4])
    bench.train_step(m, opt, batches[i % 4], None, plans=pending)
    pending = nxt
    per_step.append(time.perf_counter() - h0)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
per_step.sort()
print(f"host queues {N} steps in {t_host * 1e3:.1f} ms ({t_host / N * 1e3:.2f} ms/step; median step {per_step[N // 2] * 1e3:.2f}, "
      f"fastest {per_step[0] * 1e3:.2f}); GPU finishes them after {t_all * 1e3:.1f} ms ({t_all / N * 1e3:.2f} ms/step)")
