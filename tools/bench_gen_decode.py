"""The decode step's generator product (640 x 35000 x 512, bf16 in): time with fp32 / bf16 output, with and without
the chunk-statistics epilogue, and under the kernel-selection switches of skhip_gemm."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from specialk_amd import _lib as L

R, V, d = 640, 35000, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((R, d), device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn((V, d), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
o32 = torch.empty((R, V), dtype=torch.float32, device="cuda")
o16 = torch.empty((R, V), dtype=torch.bfloat16, device="cuda")
stats = torch.empty((R, (V + 63) // 64, 2), dtype=torch.float32, device="cuda")


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


cases = [
    ("fp32 + stats", {}, lambda: L.gemm(L.GEMM_NT, x, w, R, V, d, d, d, out=o32, ldc=V, aux=stats, epilogue=L.EPI_ROWSTATS, split_k=False)),
    ("fp32", {}, lambda: L.gemm(L.GEMM_NT, x, w, R, V, d, d, d, out=o32, ldc=V, split_k=False)),
    ("fp32 big=0", {"SKHIP_GEMM_BIG": "0"}, lambda: L.gemm(L.GEMM_NT, x, w, R, V, d, d, d, out=o32, ldc=V, split_k=False)),
    ("fp32 big=1", {"SKHIP_GEMM_BIG": "1"}, lambda: L.gemm(L.GEMM_NT, x, w, R, V, d, d, d, out=o32, ldc=V, split_k=False)),
    ("bf16 big=0", {"SKHIP_GEMM_BIG": "0"}, lambda: L.gemm(L.GEMM_NT, x, w, R, V, d, d, d, out=o16, ldc=V, split_k=False)),
    ("bf16 big=1", {"SKHIP_GEMM_BIG": "1"}, lambda: L.gemm(L.GEMM_NT, x, w, R, V, d, d, d, out=o16, ldc=V, split_k=False)),
]
for name, env, fn in cases:
    os.environ.update(env)
    t = timed(fn)
    for k in env:
        del os.environ[k]
    print(f"{name:16s} {t:8.1f} us  ({2 * R * V * d / t / 1e6:6.1f} TFLOP/s)", flush=True)
