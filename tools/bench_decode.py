"""Beam-search decode throughput at the style-transfer inference shape: Transformer-base, beam 5,
batches of 128 sentences with source lengths U[12,42], at most `--max-len` target tokens.  Weights
are random, so no sentence ends early: every batch decodes max-len steps (worst case)."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd.decode import BeamDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=2)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--beam", type=int, default=5)
ap.add_argument("--max-len", type=int, default=150)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = a.max_len
m, _ = bench.build(cfg, torch.bfloat16 if a.dtype == "bf16" else torch.float32, "cuda")
m.eval()
g = torch.Generator().manual_seed(3)
dec = BeamDecoder(m, a.beam, a.max_len)
tot_tok = tot_sent = 0
for i in range(a.batches + 1):
    s, sp = bench.make_batch(g, a.batch, 12, 42, cfg["vocab"])
    s, sp = s.cuda(), sp.cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hyps, scores = dec.run(s, sp, 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ntok = sum(len(h[0]) for h in hyps)
    print(f"batch {i}: {dt*1e3:8.1f} ms  {len(hyps)/dt:8.1f} sent/s  {ntok/dt:9.1f} tok/s (best beam)  mean len {ntok/len(hyps):.1f}", flush=True)
