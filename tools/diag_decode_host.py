import os, sys
os.environ["SK_DECODE_TIMES"] = "1"
import torch
sys.path.insert(0, "/root/repo")
import bench
from specialk_amd.decode import BeamDecoder
cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = 150
m, _ = bench.build(cfg, torch.bfloat16, "cuda")
m.eval()
g = torch.Generator().manual_seed(3)
dec = BeamDecoder(m, 5, 150)
for i in range(3):
    s, sp = bench.make_batch(g, 128, 12, 42, cfg["vocab"])
    dec.run(s.cuda(), sp.cuda(), 1)
    print({k: round(v * 1e3, 2) for k, v in dec.times.items()}, flush=True)
