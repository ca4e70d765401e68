import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from specialk_amd import ops
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg1"]
dev = torch.device("cuda", 0)
m, opt = bench.build(cfg, torch.bfloat16, dev)
ops.manual_seed(1000)
g = torch.Generator().manual_seed(7)
s, sp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
t, tp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
batch = tuple(x.to(dev) for x in (s, sp, t, tp))
for _ in range(6):
    bench.train_step(m, opt, batch, None)
torch.cuda.synchronize()
N = 50
for rep in range(3):
  acc = [0.0] * 6
  for _ in range(N):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      opt.zero_grad()
      plans = m.plan_ahead(*batch)
      t1 = time.perf_counter()
      pred, gold = m.forward_packed(*batch, plans=plans)
      t2 = time.perf_counter()
      loss, stats = ops.xent(pred, gold, True, unit_grad=True)
      t3 = time.perf_counter()
      with ops.batched_weight_grads():
          loss.backward()
      t4 = time.perf_counter()
      opt.step_and_update_lr()
      t5 = time.perf_counter()
      torch.cuda.synchronize()
      t6 = time.perf_counter()
      for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))):
          acc[i] += b - a
  print('pass', rep, 'backward', round(acc[3] / N * 1e3, 3), 'forward', round(acc[1] / N * 1e3, 3), 'total', round(sum(acc[:5]) / N * 1e3, 3))
names = ["zero_grad + plan", "forward (incl. wait for the plan's facts)", "loss", "backward", "optimiser", "drain"]
for n, a in zip(names, acc):
    print(f"{n:44s} {a / N * 1e3:7.3f} ms")
print(f"{'host total (without drain)':44s} {sum(acc[:5]) / N * 1e3:7.3f} ms")

# ---- where the backward pass spends host time: every autograd.Function's backward timed
import collections, inspect
spent, calls = collections.Counter(), collections.Counter()
for nm, cls in inspect.getmembers(ops, inspect.isclass):
    if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and "backward" in cls.__dict__:
        orig = cls.__dict__["backward"].__func__
        def make(o, n):
            def timed(*a, **k):
                t0 = time.perf_counter(); r = o(*a, **k); spent[n] += time.perf_counter() - t0; calls[n] += 1
                return r
            return timed
        cls.backward = staticmethod(make(orig, nm))
tot = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    opt.zero_grad()
    pred, gold = m.forward_packed(*batch)
    loss, stats = ops.xent(pred, gold, True, unit_grad=True)
    t3 = time.perf_counter()
    with ops.batched_weight_grads():
        loss.backward()
    tot += time.perf_counter() - t3
    opt.step_and_update_lr()
print(f"backward {tot / N * 1e3:.3f} ms per step, of which inside Function.backward bodies:")
for n, v in spent.most_common():
    print(f"  {n:16s} {calls[n] / N:5.1f} calls  {v / N * 1e3:7.3f} ms per step  ({v / calls[n] * 1e6:6.1f} us each)")
print(f"  sum {sum(spent.values()) / N * 1e3:.3f} ms")
