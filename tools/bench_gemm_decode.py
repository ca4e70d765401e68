"""Decode-step projections (640 beam rows): kernel variants of skhip_gemm on the four shapes of a decoder layer,
timed back to back (weights and activations cache-resident, as inside the decode loop)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L

shapes = [("qkv", 640, 1536, 512), ("proj", 640, 512, 512), ("ffn1", 640, 2048, 512), ("ffn2", 640, 512, 2048)]
variants = [("default", {}), ("staged", {"SKHIP_GEMM_STAGED": "1"}), ("big", {"SKHIP_GEMM_BIG": "1"})]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = []
    for vname, env in variants:
        for k in ("SKHIP_GEMM_STAGED", "SKHIP_GEMM_BIG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for sk in (False, True):
            for _ in range(5):
                L.gemm(L.GEMM_NT, A, B, M, N, K, K, K, out=out, ldc=N, bias=bias, split_k=sk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                L.gemm(L.GEMM_NT, A, B, M, N, K, K, K, out=out, ldc=N, bias=bias, split_k=sk)
            e1.record(); torch.cuda.synchronize()
            res.append(f"{vname}{'+splitk' if sk else ''} {e0.elapsed_time(e1) * 20:.1f}")
    print(f"{name:5s} {M}x{N}x{K}  us per call: " + "  ".join(res), flush=True)
