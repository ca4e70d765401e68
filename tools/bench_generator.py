"""The decode step's generator product: the vocabulary-slab kernel (skhip_generator_rowstats) against the tiled product
with statistics (skhip_gemm, SK_EPI_ROWSTATS) — same logits, same beam choices, microseconds per call.

    python3 tools/bench_generator.py [R] [V]        (GEN_PLAIN=1: row-major activations)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from specialk_amd import _lib as L  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 640
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 35000
    d, bm = 512, 5
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn(R, d, generator=g) * 1.0).to(dev, torch.bfloat16)
    w = (torch.randn(V, d, generator=g) * 0.05).to(dev, torch.bfloat16)
    cw = L.generator_chunk_cols()
    lo_a = torch.empty((R, V), dtype=torch.float32, device=dev)
    st_a = torch.empty((R, (V + 63) // 64, 2), dtype=torch.float32, device=dev)
    lo_b = torch.full((R, V), float("nan"), dtype=torch.float32, device=dev)
    st_b = torch.full((R, (V + cw - 1) // cw, 2), float("nan"), dtype=torch.float32, device=dev)

    def tiled():
        L.gemm(L.GEMM_NT, x, w, R, V, d, d, d, out=lo_a, ldc=V, aux=st_a, epilogue=L.EPI_ROWSTATS, split_k=False)

    fm = not os.environ.get("GEN_PLAIN")   # activations in the order decode_stack can write them in
    xs = L.fragment_major_rows(x) if fm else x

    def slab():
        L.generator_rowstats(xs, w, lo_b, st_b, L.ROWS_FRAGMENT_MAJOR if fm else L.ROWS_PLAIN)

    tiled()
    slab()
    torch.cuda.synchronize()
    print(f"R {R} V {V}: logits equal bit for bit: {bool(torch.equal(lo_a, lo_b))}, max |diff| {float((lo_a - lo_b).abs().max()):.3e}")
    # statistics against the logits themselves
    nch = (V + cw - 1) // cw
    pad = torch.full((R, nch * cw), float("-inf"), device=dev)
    pad[:, :V] = lo_b
    ch = pad.view(R, nch, cw)
    mx = ch.max(-1).values
    se = torch.exp(ch - mx.unsqueeze(-1)).sum(-1)
    print(f"chunk maxima equal: {bool(torch.equal(mx, st_b[..., 0]))}, sums rel err {float(((se - st_b[..., 1]).abs() / se).max()):.2e}")
    if R % bm == 0:
        n = R // bm
        scores = torch.randn(n, bm, device=dev)
        outs = []
        for lo, st, c in ((lo_a, st_a, 64), (lo_b, st_b, cw)):
            o = (torch.empty((n, bm), device=dev), torch.empty((n, bm), dtype=torch.int32, device=dev),
                 torch.empty((n, bm), dtype=torch.int32, device=dev))
            L.beam_topk_stats(lo, st, scores, n, bm, False, out=o, chunk_cols=c)
            outs.append(o)
        torch.cuda.synchronize()
        print(f"beam choices equal: prev {bool(torch.equal(outs[0][1], outs[1][1]))} tokens {bool(torch.equal(outs[0][2], outs[1][2]))}, "
              f"scores max |diff| {float((outs[0][0] - outs[1][0]).abs().max()):.2e}")
    print(f"tiled + statistics {timed(tiled):7.1f} us   slab {timed(slab):7.1f} us")


if __name__ == "__main__":
    main()
