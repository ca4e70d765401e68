#!/usr/bin/env python3
"""Diagnostic for the copy-task leg of bench.py: trains the 2+2 model exactly as the bench does, prints the loss /
accuracy trajectory, then decodes the same batches with each combination of the decode switches (one-launch stack,
chunk-statistics top-k, graph replay) and reports exact copies and mean length for each.

  python tools/diag_copy_task.py [--steps 600] [--batches 3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--batches", type=int, default=3)
    args = ap.parse_args()
    from specialk_amd import ops
    from specialk_amd.decode import BeamDecoder
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(3)
    cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = 150; cfg["n_layers"] = 2
    m, opt = bench.build(cfg, torch.bfloat16, device)
    opt.n_warmup_steps = 1500
    ops.manual_seed(5)
    for i in range(args.steps):
        s_, sp_ = bench.make_batch(g, 640, 12, 42, 2000)
        stats = bench.train_step(m, opt, tuple(x.to(device) for x in (s_, sp_, s_, sp_)), None)
        if i % 50 == 49 or i == args.steps - 1:
            st = [float(x) for x in (stats.tolist() if hasattr(stats, "tolist") else stats)]
            print(f"step {i + 1}: stats {st}", flush=True)
    m.eval()
    batches = [bench.make_batch(g, 128, 12, 42, 2000) for _ in range(args.batches)]
    for stack in ("1", "0"):
        for topk in ("1", "0"):
            for graph in ("1", "0"):
                os.environ["SK_DECODE_STACK"], os.environ["SK_DECODE_TOPK_STATS"], os.environ["SK_DECODE_GRAPH"] = stack, topk, graph
                dec = BeamDecoder(m, 5, 150)
                exact = tok = sent = 0
                for s, sp in batches:
                    hyps, _ = dec.run(s.to(device), sp.to(device), 1)
                    src = s.tolist()
                    exact += sum(1 for row, h in zip(src, hyps) if [t for t in row[1:] if t != 0] == list(h[0]))
                    tok += sum(len(h[0]) for h in hyps)
                    sent += len(hyps)
                print(f"stack={stack} topk_stats={topk} graph={graph}: exact {exact}/{sent}, mean length {tok / sent:.2f}, "
                      f"last_steps {dec.last_steps}", flush=True)


if __name__ == "__main__":
    main()
