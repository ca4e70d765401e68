"""Beam-search decode throughput at the style-transfer inference shape: Transformer-base, beam 5,
batches of 128 sentences with source lengths U[12,42], at most `--max-len` target tokens.  Weights
are random, so no sentence ends early: every batch decodes max-len steps (worst case).

--copy-task STEPS first trains the same network with this package's own training step on a synthetic
copy task (target = source, ids from a 2000-word sub-vocabulary) so that hypotheses end with EOS at
about the source length: the decode then runs the realistic number of steps per batch, and the
fraction of exactly reproduced sentences is an end-to-end check of trainer and decoder."""
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
ap.add_argument("--copy-task", type=int, default=0, metavar="STEPS")
ap.add_argument("--layers", type=int, default=6, help="layers per side (a 6+6 post-LN stack does not get off the "
                "copy task's initial plateau within a few thousand steps, in fp32 or bf16; 2+2 learns it in 300)")
a = ap.parse_args()
cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = a.max_len; cfg["n_layers"] = a.layers
m, opt = bench.build(cfg, torch.bfloat16 if a.dtype == "bf16" else torch.float32, "cuda")
g = torch.Generator().manual_seed(3)
vocab_used = cfg["vocab"]
if a.copy_task:
    from specialk_amd import ops
    vocab_used = 2000
    opt.n_warmup_steps = int(os.environ.get("SK_COPY_WARMUP", "1500"))
    ops.manual_seed(5)
    t0 = time.perf_counter()
    for it in range(a.copy_task):
        s_, sp_ = bench.make_batch(g, 640, 12, 42, vocab_used)
        batch = tuple(x.cuda() for x in (s_, sp_, s_, sp_))
        stats = bench.train_step(m, opt, batch, None)
        if (it + 1) % 250 == 0:
            st = stats.tolist()
            print(f"copy task step {it + 1}: loss/token {st[0] / st[2]:.3f}  accuracy {st[1] / st[2]:.3f}  "
                  f"({time.perf_counter() - t0:.1f} s)", flush=True)
m.eval()
dec = BeamDecoder(m, a.beam, a.max_len)
tot_tok = tot_sent = exact = 0
tot_dt = 0.0
for i in range(a.batches + 1):
    s, sp = bench.make_batch(g, a.batch, 12, 42, vocab_used)
    s, sp = s.cuda(), sp.cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hyps, scores = dec.run(s, sp, 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ntok = sum(len(h[0]) for h in hyps)
    if i > 0:  # batch 0 is the warm-up
        tot_tok += ntok; tot_sent += len(hyps); tot_dt += dt
        src = s.cpu().tolist()
        exact += sum(1 for row, h in zip(src, hyps) if [t for t in row[1:] if t != 0] == list(h[0]))
    if a.batches > 8 and 0 < i < a.batches:
        continue
    print(f"batch {i}: {dt*1e3:8.1f} ms  {len(hyps)/dt:8.1f} sent/s  {ntok/dt:9.1f} tok/s (best beam)  mean len {ntok/len(hyps):.1f}", flush=True)
print(f"total: {tot_sent} sentences in {tot_dt:.2f} s = {tot_sent / tot_dt:.1f} sentences/s, {tot_tok / tot_dt:.0f} tokens/s, "
      f"mean output length {tot_tok / max(tot_sent, 1):.1f}, exact copies {exact}/{tot_sent}", flush=True)
