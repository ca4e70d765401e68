#!/usr/bin/env python3
"""LayerNorm forward / backward (skhip_add_ln_fwd, skhip_add_ln_bwd incl. its parameter reduction) on the training
shape (25 639 rows of 512, bf16, dropout 0.1): microseconds per call and the rate of the tensors a call has to move
(forward 3, backward 5 row tensors), operands rotated between calls.   python tools/bench_ln.py [--iters 40]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from specialk_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rows", type=int, default=25639)
    args = ap.parse_args()
    T, d, NB = args.rows, 512, 4
    dev, bf = "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(0)
    x = [(torch.randn(T, d, generator=g)).to(bf).to(dev) for _ in range(NB)]
    res = [(torch.randn(T, d, generator=g)).to(bf).to(dev) for _ in range(NB)]
    dy = [(torch.randn(T, d, generator=g) * 0.1).to(bf).to(dev) for _ in range(NB)]
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    stats = [L.add_ln_fwd(x[i], res[i], gamma, beta, None, 1e-5, 0.1, 7)[1:] for i in range(NB)]
    big = torch.randn(8192, 8192, device=dev, dtype=bf)

    def timed(fn):
        for i in range(NB):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            torch.mm(big, big)
        e0.record()
        for i in range(args.iters):
            fn(i % NB)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.iters

    tf = timed(lambda i: L.add_ln_fwd(x[i], res[i], gamma, beta, None, 1e-5, 0.1, 7))
    tb = timed(lambda i: L.add_ln_bwd(dy[i], x[i], res[i], gamma, None, stats[i][0], stats[i][1], 0.1, 7, want_dx_colsum=True))
    row = T * d * 2
    print(f"rows {T}: forward {tf:6.1f} us ({3 * row / tf / 1e6:4.2f} TB/s)   backward + parameter reduction {tb:6.1f} us ({5 * row / tb / 1e6:4.2f} TB/s)")


if __name__ == "__main__":
    main()
