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
import gc
ev = []
def cb(phase, info):
    ev.append((phase, info['generation'], time.perf_counter()))
gc.callbacks.append(cb)
ts = []
for i in range(120):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.train_step(m, opt, batch, None)
    t1 = time.perf_counter()
    ts.append((t1 - t0) * 1e3)
print("host ms per step:", " ".join(f"{x:.2f}" for x in ts))
st = None
for ph, gen, t in ev:
    if ph == "start": st = t
    elif gen == 2: print(f"gen2 collection {1e3 * (t - st):.1f} ms")
print("allocator:", torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"])
