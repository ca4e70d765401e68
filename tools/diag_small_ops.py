"""Which lines of the package issue the small copy / fill / reduce launches of a training step: one step under
torch.profiler with stacks, every aten op that launches a kernel attributed to the innermost frame inside this
repository.  Usage: python tools/diag_small_ops.py [cfg2]"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import ops

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = torch.device("cuda", 0)
m, opt = bench.build(cfg, torch.bfloat16, dev)
ops.manual_seed(1000)
g = torch.Generator().manual_seed(7)
s, sp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
t, tp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
batch = tuple(x.to(dev) for x in (s, sp, t, tp))
for _ in range(3):
    bench.train_step(m, opt, batch, None)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.train_step(m, opt, batch, None)
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    kern = [k for k in ev.kernels]
    if not kern:
        continue
    where = "?"
    for fr in ev.stack or []:
        if ROOT in fr and "tools/" not in fr:
            where = fr.replace(ROOT + "/", "")
            break
    key = (ev.name, where)
    agg[key] += len(kern)
    tim[key] += sum(k.duration for k in kern)
for key, n in sorted(agg.items(), key=lambda kv: -tim[kv[0]]):
    print(f"{n:4d} launches {tim[key]:8.1f} us  {key[0]:28s} {key[1]}")
