"""Timing of a row-split NT GEMM: the first rows (whole rounds of 256-tiles) on the 256-tile kernel,
the remaining rows on the 128-tile kernel, against either kernel alone (GPU box)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for M, N, K in [(25472, 2048, 512), (25472, 1536, 512), (26200, 2048, 512), (26200, 1024, 512), (25472, 512, 2048)]:
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn((N, K), device=dev, generator=g).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    tn = (N + 255) // 256
    rounds = ((M + 255) // 256 * tn) // 256
    mb = (rounds * 256 // tn) * 256
    def run(force, m0=0, m1=M):
        os.environ["SKHIP_GEMM_BIG"] = force
        L.gemm(0, A[m0:m1], B, m1 - m0, N, K, K, K, out=out[m0:m1], ldc=N, bias=bias, epilogue=L.EPI_RELU, split_k=False)
    t_small = timeit(lambda: run("0"))
    t_big = timeit(lambda: run("1"))
    def mixed():
        run("1", 0, mb)
        if mb < M: run("0", mb, M)
    t_mix = timeit(mixed)
    t_bigpart = timeit(lambda: run("1", 0, mb))
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: small {t_small:6.1f} us ({fl/t_small/1e6:5.0f} TF) | big {t_big:6.1f} | mixed {t_mix:6.1f} ({fl/t_mix/1e6:5.0f} TF) [big rows 0..{mb}: {t_bigpart:6.1f}]", flush=True)
