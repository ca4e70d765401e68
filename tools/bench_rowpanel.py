#!/usr/bin/env python3
"""Row-panel products against the tiled kernels on the training shapes (T = 25 639 packed rows, d = 512, F = 2048):
microseconds per call, events on the launch stream, the operands of the different calls rotated so that no call
finds its activations in the caches of the call before it.

  python tools/bench_rowpanel.py [--rows 25639] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from specialk_amd import _lib as L, ops  # noqa: E402


_big = None


def timed(fn, iters, nbuf):
    """GPU time per call: a few milliseconds of other work are queued first, so that the host has enqueued all the
    calls before the first of them starts (a 20 us kernel is otherwise timed at the host's 25 us per launch)."""
    global _big
    if _big is None:
        _big = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    for i in range(3):
        fn(i % nbuf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4):
        torch.mm(_big, _big)
    e0.record()
    for i in range(iters):
        fn(i % nbuf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=25639)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--vendor", action="store_true", help="add the vendor library (torch.mm: hipBLASLt) on the plain products")
    args = ap.parse_args()
    T, d, F, NB = args.rows, 512, 2048, 4
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    bf = torch.bfloat16

    def act(n):
        return [(torch.randn(T, n, generator=g) * 0.5).to(bf).to(dev) for _ in range(NB)]

    def par(n, k):
        return torch.nn.Parameter((torch.randn(n, k, generator=g) * k ** -0.5).to(dev))

    x512 = act(512)
    out512 = [torch.empty(T, 512, dtype=bf, device=dev) for _ in range(NB)]
    out2048 = [torch.empty(T, 2048, dtype=bf, device=dev) for _ in range(NB)]
    out1536 = [torch.empty(T, 1536, dtype=bf, device=dev) for _ in range(NB)]
    signs = [torch.empty(T, F // 16, dtype=torch.int16, device=dev) for _ in range(NB)]
    pats = [L.rowpanel_pattern(T, F, dev) for _ in range(NB)]
    W1, W2, Wo, Wqkv = par(F, d), par(d, F), par(d, d), par(1536, d)
    b1, b2, b3 = torch.zeros(F, device=dev), torch.zeros(d, device=dev), torch.zeros(1536, device=dev)
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    c = {k: ops.compute_weight(v, bf) for k, v in dict(W1=W1, W2=W2, Wo=Wo, Wqkv=Wqkv).items()}
    W2t = ops.derived_weight("catT", [W2])   # (F, d)^T ... K-contiguous operand of dz W2
    f = dict(W1=ops.derived_weight("fm", [W1]), W2=ops.derived_weight("fm", [W2]), Wo=ops.derived_weight("fm", [Wo]),
             Wqkv=ops.derived_weight("fm", [Wqkv]), W2T=ops.derived_weight("fmT", [W2]))
    RB = L.EPI_RELU | L.EPI_RELU_BITS_OUT
    rows = []

    def add(name, flop, tiled, panel):
        t0 = timed(tiled, args.iters, NB)
        t1 = timed(panel, args.iters, NB)
        rows.append((name, t0, t1, flop / t0 / 1e6, flop / t1 / 1e6))

    fl = lambda n, k: 2.0 * T * n * k
    add("w_1: N 2048 K 512 bias relu pattern", fl(F, d),
        lambda i: L.gemm(L.GEMM_NT, x512[i], c["W1"], T, F, d, d, d, out=out2048[i], ldc=F, bias=b1, aux=signs[i], epilogue=RB),
        lambda i: L.rowpanel_gemm(x512[i], f["W1"], F, bias=b1, out=out2048[i], ldc=F, signs=pats[i], epilogue=RB))
    add("dz w_2: N 2048 K 512 masked", fl(F, d),
        lambda i: L.gemm(L.GEMM_NT, x512[i], W2t, T, F, d, d, d, out=out2048[i], ldc=F, aux=signs[i], epilogue=L.EPI_RELU_MASK_BITS),
        lambda i: L.rowpanel_gemm(x512[i], f["W2T"], F, out=out2048[i], ldc=F, signs=pats[i], epilogue=L.EPI_RELU_MASK_BITS))
    add("q|k|v: N 1536 K 512 bias", fl(1536, d),
        lambda i: L.gemm(L.GEMM_NT, x512[i], c["Wqkv"], T, 1536, d, d, d, out=out1536[i], ldc=1536, bias=b3),
        lambda i: L.rowpanel_gemm(x512[i], f["Wqkv"], 1536, bias=b3, out=out1536[i], ldc=1536))
    add("N 512 K 512 bias", fl(d, d),
        lambda i: L.gemm(L.GEMM_NT, x512[i], c["Wo"], T, d, d, d, d, out=out512[i], ldc=d, bias=b2),
        lambda i: L.rowpanel_gemm(x512[i], f["Wo"], d, bias=b2, out=out512[i], ldc=d))
    add("N 512 K 512 accumulate", fl(d, d),
        lambda i: L.gemm(L.GEMM_NT, x512[i], c["Wo"], T, d, d, d, d, out=out512[i], ldc=d, epilogue=L.EPI_ACCUM),
        lambda i: L.rowpanel_gemm(x512[i], f["Wo"], d, out=out512[i], ldc=d, epilogue=L.EPI_ACCUM))

    def tiled_ln(i, x, w, k):
        sub = L.gemm(L.GEMM_NT, x[i], w, T, d, k, k, k, out=out512[i], ldc=d, bias=b2)
        L.add_ln_fwd(sub, x512[(i + 1) % NB], gamma, beta, None, 1e-5, 0.1, 1234)

    add("fc + LayerNorm: N 512 K 512", fl(d, d), lambda i: tiled_ln(i, x512, c["Wo"], d),
        lambda i: L.rowpanel_gemm_ln(x512[i], f["Wo"], b2, x512[(i + 1) % NB], gamma, beta, None, 1e-5, 0.1, 1234))
    Wkv = par(6144, d)
    b4 = torch.zeros(6144, device=dev)
    out6144 = [torch.empty(T, 6144, dtype=bf, device=dev) for _ in range(2)]
    ckv, fkv = ops.compute_weight(Wkv, bf), ops.derived_weight("fm", [Wkv])
    add("k|v of 6 layers: N 6144 K 512 bias", fl(6144, d),
        lambda i: L.gemm(L.GEMM_NT, x512[i], ckv, T, 6144, d, d, d, out=out6144[i % 2], ldc=6144, bias=b4),
        lambda i: L.rowpanel_gemm(x512[i], fkv, 6144, bias=b4, out=out6144[i % 2], ldc=6144))

    if args.vendor:   # the plain products (no epilogue beyond the cast) through torch.mm, same operands and rotation
        print(f"rows {T}: vendor library (torch.mm) against skhip_gemm / skhip_rowpanel_gemm, no bias; microseconds (TFLOP/s)")
        for N, K in ((512, 2048), (2048, 512), (512, 512), (512, 1536), (1536, 512), (35000, 512)):
            xs = act(K)
            W = par(N, K)
            Wb, Wf = ops.compute_weight(W, bf), (ops.derived_weight("fm", [W]) if K == 512 and N % 64 == 0 and N <= 6144 else None)
            outs = [torch.empty(T, N, dtype=bf, device=dev) for _ in range(2 if N > 4096 else NB)]
            tv = timed(lambda i: torch.mm(xs[i], Wb.t(), out=outs[i % len(outs)]), args.iters, NB)
            tt = timed(lambda i: L.gemm(L.GEMM_NT, xs[i], Wb, T, N, K, K, K, out=outs[i % len(outs)], ldc=N), args.iters, NB)
            tp = timed(lambda i: L.rowpanel_gemm(xs[i], Wf, N, out=outs[i % len(outs)], ldc=N), args.iters, NB) if Wf is not None else None
            f = fl(N, K)
            line = f"N {N:6d} K {K:5d}: vendor {tv:8.1f} ({f / tv / 1e6:5.0f})  tiled {tt:8.1f} ({f / tt / 1e6:5.0f})"
            if tp is not None:
                line += f"  row panel {tp:8.1f} ({f / tp / 1e6:5.0f})"
            print(line)
        print()
    print(f"rows {T}: microseconds per call (TFLOP/s)")
    print(f"{'product':44s} {'tiled':>16s} {'row panel':>16s}")
    for name, t0, t1, r0, r1 in rows:
        print(f"{name:44s} {t0:8.1f} ({r0:5.0f}) {t1:8.1f} ({r1:5.0f})")


if __name__ == "__main__":
    main()
