"""GEMM micro-benchmark (GPU box): the cfg-2 training shapes through skhip_gemm, HIP-event timed."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L

T, d, F, V = 33280, 512, 2048, 35000
SHAPES = [  # name, layout, M, N, K, out_dtype
    ("qkv/fc fwd  NT T x512 x512", 0, T, d, d, torch.bfloat16),
    ("ffn1 fwd    NT T x2048x512", 0, T, F, d, torch.bfloat16),
    ("ffn2 fwd    NT T x512 x2048", 0, T, d, F, torch.bfloat16),
    ("gen fwd     NT T x35000x512", 0, T, V, d, torch.bfloat16),
    ("dgrad       NN T x512 x512", 1, T, d, d, torch.bfloat16),
    ("ffn2 dgrad  NN T x2048x512", 1, T, F, d, torch.bfloat16),
    ("ffn1 dgrad  NN T x512 x2048", 1, T, d, F, torch.bfloat16),
    ("gen dgrad   NN T x512 x35000", 1, T, d, V, torch.bfloat16),
    ("wgrad       TN 512 x512 xT", 2, d, d, T, torch.float32),
    ("ffn wgrad   TN 2048x512 xT", 2, F, d, T, torch.float32),
    ("ffn wgrad   TN 512 x2048xT", 2, d, F, T, torch.float32),
    ("gen wgrad   TN 35000x512xT", 2, V, d, T, torch.float32),
]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for name, layout, M, N, K, odt in SHAPES:
    A = torch.randn((K, M) if layout == 2 else (M, K), device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn((N, K) if layout == 0 else (K, N), device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=odt, device=dev)
    run = lambda: L.gemm(layout, A, B, M, N, K, A.shape[1], B.shape[1], out=out, ldc=N)
    for _ in range(3): run()
    torch.cuda.synchronize()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * M * N * K
    byts = A.numel() * 2 + B.numel() * 2 + out.numel() * out.element_size()
    print(f"{name:32s} {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s   min-traffic {byts / us / 1e3:7.1f} GB/s")
