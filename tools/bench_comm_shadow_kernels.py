"""Single kernels under a CU-holding stand-in kernel (see bench_comm_shadow.py): time of a launch with n CUs held,
with and without leaving them out of the launch's planning (skhip_set_reserved_cus).
Usage: python tools/bench_comm_shadow_kernels.py [blocks ...]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specialk_amd import _lib as L

occ = ctypes.CDLL(os.path.join(ROOT, "tools", "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
blocks_list = [int(x) for x in sys.argv[1:]] or [8, 32]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream()
sink = torch.zeros(1, dtype=torch.int32, device=dev)
M = 25639


def rnd(*shape):
    return torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)


cases = []
for N, K in ((512, 512), (512, 2048), (2048, 512), (1536, 512)):
    A, B, out = rnd(M, K), rnd(N, K), torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    cases.append((f"gemm NT {M}x{N}x{K}", lambda A=A, B=B, out=out, N=N, K=K: L.gemm(0, A, B, M, N, K, K, K, out=out, ldc=N)))
x, res = rnd(M, 512), rnd(M, 512)
gamma, beta = torch.ones(512, device=dev), torch.zeros(512, device=dev)
cases.append(("add_ln_fwd", lambda: L.add_ln_fwd(x, res, gamma, beta, None, 1e-5, 0.1, 7)))
logits = rnd(M, 35000)
gold = torch.randint(1, 35000, (M,), device=dev)
stats = torch.zeros(3, dtype=torch.float64, device=dev)
if hasattr(L, "xent_fwd_bwd"):
    cases.append(("xent", lambda: L.xent_fwd_bwd(logits, gold, logits, stats, 1.0, 0.1, 0)))


def timed(fn, n_occ, reserve, reps=6):
    L.set_reserved_cus(reserve)
    torch.cuda.synchronize()
    if n_occ:
        for _ in range(8):
            assert occ.occupy(n_occ, 1000, 24576, sink.data_ptr(), side.cuda_stream) == 0
    torch.cuda._sleep(200000)  # let the stand-in take its CUs first (the same wait when nothing is held)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print(f"{'kernel':30s} {'free us':>9s} " + " ".join(f"{'held ' + str(n):>9s} {'+reserve':>9s}" for n in blocks_list))
for name, fn in cases:
    fn(); torch.cuda.synchronize()
    row = [sorted(timed(fn, 0, 0) for _ in range(3))[1]]
    for n in blocks_list:
        row.append(sorted(timed(fn, n, 0) for _ in range(3))[1])
        row.append(sorted(timed(fn, n, n) for _ in range(3))[1])
    print(f"{name:30s} " + " ".join(f"{t:9.1f}" for t in row), flush=True)
L.set_reserved_cus(0)
