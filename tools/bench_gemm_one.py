"""One GEMM shape, many launches (for rocprofv3 --pmc runs).  Usage: bench_gemm_one.py layout M N K [out_fp32]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L
layout, M, N, K = (int(x) for x in sys.argv[1:5])
odt = torch.float32 if len(sys.argv) > 5 else torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn((K, M) if layout == 2 else (M, K), device="cuda", generator=g).to(torch.bfloat16)
B = torch.randn((N, K) if layout == 0 else (K, N), device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty((M, N), dtype=odt, device="cuda")
for _ in range(5):
    L.gemm(layout, A, B, M, N, K, A.shape[1], B.shape[1], out=out, ldc=N)
torch.cuda.synchronize()
