"""A/B of the bf16 GEMM kernels on the Transformer-base training shapes, interleaved rounds in
one process (GPU box).  Variants are selected through the library's environment switches.
Also checks each variant against torch.matmul in fp32 on the same operands."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L

T, d, F, V = 33280, 512, 2048, 35000
P = 25472
SHAPES = [
    ("ffnw TN 2048x512xP", 2, F, d, P, torch.float32),
    ("ffnw TN 512x2048xP", 2, d, F, P, torch.float32),
    ("qkvw TN 1536x512xP", 2, 3 * d, d, P, torch.float32),
    ("kvw TN 1024x512xP", 2, 2 * d, d, P, torch.float32),
    ("ow TN 512x512xP", 2, d, d, P, torch.float32),
    ("gen NT Px35000x512", 0, P, V, d, torch.bfloat16),
    ("gendx NT Px512x35000", 0, P, d, V, torch.bfloat16),
    ("genw TN 35000x512xP", 2, V, d, P, torch.float32),
    ("ffn2 NT Px512x2048", 0, P, d, F, torch.bfloat16),
    ("qkv NT Px1536x512", 0, P, 3 * d, d, torch.bfloat16),
    ("fwd  NT T x512 x512", 0, T, d, d, torch.bfloat16),
    ("ffn1 NT T x2048x512", 0, T, F, d, torch.bfloat16),
    ("ffn2 NT T x512 x2048", 0, T, d, F, torch.bfloat16),
    ("gen  NT T x35000x512", 0, T, V, d, torch.bfloat16),
    ("packed NT 25472x512x512", 0, 25472, d, d, torch.bfloat16),
    ("packed NT 25472x2048x512", 0, 25472, F, d, torch.bfloat16),
    ("packed NT 25472x1536x512", 0, 25472, 3 * d, d, torch.bfloat16),
    ("packed NT 25472x1024x512", 0, 25472, 2 * d, d, torch.bfloat16),
    ("packed NT 25472x512x1536", 0, 25472, d, 3 * d, torch.bfloat16),
    ("cube NT 8192x8192x8192", 0, 8192, 8192, 8192, torch.bfloat16),
    ("cube NT 4096x4096x4096", 0, 4096, 4096, 4096, torch.bfloat16),
    ("dgrad NN T x512 x512", 1, T, d, d, torch.bfloat16),
    ("dgrad NN T x2048x512", 1, T, F, d, torch.bfloat16),
    ("dgrad NN T x512 x2048", 1, T, d, F, torch.bfloat16),
    ("wgrad TN 512 x512 xT", 2, d, d, T, torch.float32),
    ("wgrad TN 2048x512 xT", 2, F, d, T, torch.float32),
    ("wgrad TN 512 x2048xT", 2, d, F, T, torch.float32),
]
VARIANTS = [("small g1", {"SKHIP_GEMM_BIG": "0", "SKHIP_GEMM_GROUP": "1"}), ("small g8", {"SKHIP_GEMM_BIG": "0", "SKHIP_GEMM_GROUP": "8"}),
            ("big g1", {"SKHIP_GEMM_BIG": "1", "SKHIP_GEMM_GROUP": "1"}), ("big g8", {"SKHIP_GEMM_BIG": "1", "SKHIP_GEMM_GROUP": "8"})]
# SK_AB_VARIANTS="name:ENV=1,ENV2=0;name2:-" replaces the variant list
if os.environ.get("SK_AB_VARIANTS"):
    VARIANTS = []
    for a in os.environ["SK_AB_VARIANTS"].split(";"):
        vname, _, envs = a.partition(":")
        VARIANTS.append((vname, dict(kv.split("=") for kv in envs.split(",") if kv and kv != "-")))
_keys = sorted({k for _, e in VARIANTS for k in e})
only = sys.argv[1] if len(sys.argv) > 1 else ""
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for name, layout, M, N, K, odt in SHAPES:
    if only and only not in name:
        continue
    A = torch.randn((K, M) if layout == 2 else (M, K), device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn((N, K) if layout == 0 else (K, N), device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=odt, device=dev)
    run = lambda: L.gemm(layout, A, B, M, N, K, A.shape[1], B.shape[1], out=out, ldc=N)
    rows = slice(0, min(M, 1024))
    a32 = (A.float().t() if layout == 2 else A.float())
    b32 = (B.float().t() if layout == 0 else B.float())
    ref_head = a32[rows] @ b32[:, :2048]
    ref_tail = a32[-300:] @ b32[:, -300:]
    times = {v: [] for v, _ in VARIANTS}
    errs = {}
    for rnd in range(5):
        for v, env in VARIANTS:
            for k in _keys:
                os.environ.pop(k, None)
            os.environ.update(env)
            if rnd == 0:
                out.zero_()
                run()
                torch.cuda.synchronize()
                e1 = (out[rows, :2048].float() - ref_head).abs().max().item() / ref_head.abs().max().item()
                e2 = (out[-300:, -300:].float() - ref_tail).abs().max().item() / ref_tail.abs().max().item()
                errs[v] = max(e1, e2)
            e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1_.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1_) * 1e3 / 5)
    fl = 2.0 * M * N * K
    line = f"{name:28s}"
    for v, _ in VARIANTS:
        t = sorted(times[v])
        line += f" | {v}: {t[2]:7.1f} us {fl / t[2] / 1e6:6.0f} TF e{errs[v]:.0e}"
    print(line, flush=True)
