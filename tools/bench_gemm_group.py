"""Rasterisation group (tile rows walked before moving to the next tile column) of the 256-tile kernel on the
generator forward product 25.6 k x 35 000 x 512: interleaved rounds, one process.  Usage: bench_gemm_group.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L

shapes = [("gen fwd", 25639, 35000, 512), ("gen fwd cfg4", 12800, 44544, 1024), ("ffn1", 25639, 2048, 512), ("qkv", 25639, 1536, 512)]
groups = [1, 2, 4, 8, 16, 32]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    best = {}
    for rnd in range(4):
        for grp in groups:
            os.environ["SKHIP_GEMM_GROUP"] = str(grp)
            os.environ["SKHIP_GEMM_BIG"] = "1"
            for _ in range(2):
                L.gemm(L.GEMM_NT, A, B, M, N, K, K, K, out=out, ldc=N)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.gemm(L.GEMM_NT, A, B, M, N, K, K, K, out=out, ldc=N)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 200
            best[grp] = min(best.get(grp, 1e9), us)
    print(name, M, N, K, "  ".join(f"g{grp}: {us:7.1f} us ({2 * M * N * K / us / 1e6:5.0f} TF)" for grp, us in best.items()), flush=True)
