"""NT against NN (row-contiguous second operand) on the 256-tile kernel, for the input-gradient shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L
dev = "cuda"
for (M, N, K) in [(25472, 512, 35000), (25472, 512, 2048), (25472, 512, 1536), (25472, 512, 1024), (25472, 512, 512), (25472, 2048, 512)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    Bt = torch.randn(N, K, device=dev).bfloat16()
    Bn = Bt.t().contiguous()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    line = f"{M}x{N}x{K}:"
    for name, env, fn in [("NT big", "1", lambda: L.gemm(L.GEMM_NT, A, Bt, M, N, K, K, K, out=out, ldc=N, split_k=False)),
                          ("NT auto", None, lambda: L.gemm(L.GEMM_NT, A, Bt, M, N, K, K, K, out=out, ldc=N, split_k=False)),
                          ("NN big", "1", lambda: L.gemm(L.GEMM_NN, A, Bn, M, N, K, K, N, out=out, ldc=N, split_k=False)),
                          ("NN auto", None, lambda: L.gemm(L.GEMM_NN, A, Bn, M, N, K, K, N, out=out, ldc=N, split_k=False))]:
        if env is None: os.environ.pop("SKHIP_GEMM_BIG", None)
        else: os.environ["SKHIP_GEMM_BIG"] = env
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 3)
        ts.sort()
        line += f"  {name} {ts[4]:7.1f} us"
    print(line, flush=True)
