"""A/B of the 256-tile bf16 NT kernels on the training step's short-K shapes with their real epilogues:
one-tile-per-block launches against the persistent K-step-stream form, interleaved rounds in one process.
Each variant is checked against an fp32 torch product on sampled rows.  "hot" = five launches back to back
(operands in the caches), "cold" = one launch after 600 MB of unrelated stores."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L

EPI = {"bias": L.EPI_BIAS, "relu+bits": L.EPI_BIAS | L.EPI_RELU | L.EPI_RELU_BITS_OUT, "accum": L.EPI_ACCUM,
       "maskbits": L.EPI_RELU_MASK_BITS, "none": 0}
P, Q = 25639, 25970
SHAPES = [(P, 512, 512, "bias"), (Q, 512, 512, "accum"), (Q, 1024, 512, "bias"), (P, 1536, 512, "bias"),
          (P, 2048, 512, "relu+bits"), (Q, 2048, 512, "maskbits"), (P, 512, 2048, "bias"), (Q, 512, 1536, "accum"),
          (P, 35000, 512, "none"), (8192, 8192, 8192, "none")]
VARIANTS = [("big", {"SKHIP_GEMM_PERSIST": "0", "SKHIP_GEMM_BIG": "1", "SKHIP_GEMM_STORE": "0"}),
            ("big sc1", {"SKHIP_GEMM_PERSIST": "0", "SKHIP_GEMM_BIG": "1", "SKHIP_GEMM_STORE": "1"}),
            ("big nt", {"SKHIP_GEMM_PERSIST": "0", "SKHIP_GEMM_BIG": "1", "SKHIP_GEMM_STORE": "2"}),
            ("stream sc1", {"SKHIP_GEMM_PERSIST": "1", "SKHIP_GEMM_BIG": "1", "SKHIP_GEMM_STORE": "1"})]
only = sys.argv[1] if len(sys.argv) > 1 else ""
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
junk = torch.empty(600 << 20, dtype=torch.uint8, device=dev)


def setenv(env):
    for k, v in env.items():
        if v == "":
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


for M, N, K, epi in SHAPES:
    name = f"{M}x{N}x{K} {epi}"
    if only and only not in name:
        continue
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    B = (torch.randn((N, K), device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    C0 = torch.randn((M, N), device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    signs = torch.empty((M, N // 16), dtype=torch.int16, device=dev) if "bits" in epi else None
    if epi == "maskbits":
        signs.random_(-32768, 32767, generator=g)
    flags = EPI[epi]

    def run():
        return L.gemm(L.GEMM_NT, A, B, M, N, K, K, K, out=out, ldc=N, bias=bias if flags & L.EPI_BIAS else None,
                      aux=signs, epilogue=flags & ~L.EPI_BIAS)

    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 300), torch.arange(M - 300, M)]).to(dev)
    cols = slice(0, min(N, 4096))
    ref = A[rows].float() @ B.float().t()[:, cols]
    if flags & L.EPI_BIAS:
        ref += bias[cols]
    if flags & L.EPI_RELU:
        ref.clamp_(min=0)
    if flags & L.EPI_RELU_MASK_BITS:
        w = signs[rows].int() & 0xFFFF
        m = ((w.unsqueeze(-1) >> torch.arange(16, device=dev)) & 1).reshape(len(rows), -1)[:, cols]
        ref = ref * m
    if flags & L.EPI_ACCUM:
        ref += C0[rows][:, cols].float()
    times = {v: {"hot": [], "cold": []} for v, _ in VARIANTS}
    errs = {}
    for rnd in range(5):
        for v, env in VARIANTS:
            setenv(env)
            if rnd == 0:
                out.copy_(C0)
                run()
                torch.cuda.synchronize()
                errs[v] = ((out[rows][:, cols].float() - ref).abs().max() / ref.abs().max()).item()
                if flags & L.EPI_RELU_BITS_OUT:
                    w = signs[rows].int() & 0xFFFF
                    m = ((w.unsqueeze(-1) >> torch.arange(16, device=dev)) & 1).reshape(len(rows), -1)[:, cols]
                    assert torch.equal(m.bool(), out[rows][:, cols] > 0), (name, v, "sign pattern")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[v]["hot"].append(e0.elapsed_time(e1) * 1e3 / 5)
            junk.fill_(rnd)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            times[v]["cold"].append(e0.elapsed_time(e1) * 1e3)
    fl = 2.0 * M * N * K
    line = f"{name:28s}"
    for v, _ in VARIANTS:
        h, c = sorted(times[v]["hot"])[2], sorted(times[v]["cold"])[2]
        line += f" | {v}: hot {h:7.1f} us {fl / h / 1e6:5.0f} TF  cold {c:7.1f}  e{errs[v]:.0e}"
    print(line, flush=True)
