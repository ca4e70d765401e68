"""A/B of library switches on the worst-case decode leg (6+6 model, 640 beam rows, `steps` steps per batch), interleaved
rounds in one process.  Usage: python tools/bench_decode_ab.py "name:ENV=1,ENV2=0" "name2:-" ... [--steps N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from specialk_amd.decode import BeamDecoder

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 150
args = [a for a in args if a != str(steps) or "--steps" not in sys.argv]
variants = []
for a in args:
    name, _, envs = a.partition(":")
    variants.append((name, dict(kv.split("=") for kv in envs.split(",") if kv and kv != "-")))
keys = sorted({k for _, e in variants for k in e})
cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = 150
m, _ = bench.build(cfg, torch.bfloat16, "cuda")
m.eval()
g = torch.Generator().manual_seed(3)
batches = [tuple(x.cuda() for x in bench.make_batch(g, 128, 12, 42, cfg["vocab"])) for _ in range(2)]
times = {n: [] for n, _ in variants}
for rnd in range(4):
    for name, env in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        dec = BeamDecoder(m, 5, steps)
        if rnd == 0:
            dec.run(*batches[0], 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            dec.run(*b, 1)
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / (len(batches) * steps) * 1e3)
for name, _ in variants:
    t = sorted(times[name])
    print(f"{name:20s} median {t[len(t) // 2]:.4f} ms/step  min {t[0]:.4f}  max {t[-1]:.4f}", flush=True)
