"""GEMM shapes timed one launch at a time with the caches in a stated condition: "warm" = the same operands as the
launch before, "cold" = a 1 GiB copy between launches (operands come from HBM, as after an unrelated kernel),
"fresh" = the A operand was just written by an elementwise kernel (as after LayerNorm in the training step).
Usage: python tools/bench_gemm_ctx.py [M ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L

Ms = [int(x) for x in sys.argv[1:]] or [25639, 25970, 25600, 26112]
NK = [(512, 512), (2048, 512), (512, 2048), (1536, 512), (512, 1536), (1024, 512), (512, 1024)]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
junk_a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
junk_b = torch.empty(1 << 28, dtype=torch.float32, device=dev)
print(f"{'M':>6s} {'N':>5s} {'K':>5s} | {'warm us':>8s} {'TF':>5s} | {'cold us':>8s} {'TF':>5s} | {'fresh us':>8s} {'TF':>5s}")
for M in Ms:
    for N, K in NK:
        A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
        A2 = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
        B = torch.randn((N, K), device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        res = []
        for mode in ("warm", "cold", "fresh"):
            ts = []
            for it in range(8):
                if mode == "cold":
                    junk_b.copy_(junk_a)
                elif mode == "fresh":
                    junk_b.copy_(junk_a)
                    A.copy_(A2)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.gemm(0, A, B, M, N, K, K, K, out=out, ldc=N)
                e1.record()
                torch.cuda.synchronize()
                if it >= 2:
                    ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res.append(ts[len(ts) // 2])
        fl = 2.0 * M * N * K
        print(f"{M:6d} {N:5d} {K:5d} | " + " | ".join(f"{t:8.1f} {fl / t / 1e6:5.0f}" for t in res), flush=True)
