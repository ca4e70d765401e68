"""Where a decode call's set-up goes: encoder pass, encoder-side keys / values, cache allocation (wall clock with a
device synchronisation after each part, cfg-2 model, 128 sentences)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from specialk_amd import _lib as L, ops
cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = 150
m, _ = bench.build(cfg, torch.bfloat16, "cuda")
m.eval()
g = torch.Generator().manual_seed(3)
dev = torch.device("cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - a) / n
    torch.cuda.synchronize()
    return host * 1e3, (time.perf_counter() - a) / n * 1e3
s, sp = bench.make_batch(g, 128, 12, 42, cfg["vocab"])
s, sp = s.cuda(), sp.cuda()
with torch.no_grad():
    print("encoder pass (host ms, host+device ms):", t(lambda: m.encoder(s, sp)))
    enc, *_ = m.encoder(s, sp)
    n, Ls = s.shape
    enc2d = enc.reshape(n * Ls, 512)
    layers = list(m.decoder.layer_stack)
    def kv():
        K = torch.empty((6, n, Ls, 512), dtype=torch.bfloat16, device=dev)
        for li, l in enumerate(layers):
            a = l.enc_attn
            for w in (a.w_ks, a.w_vs):
                ww = ops.compute_weight(w.weight, torch.bfloat16)
                L.gemm(L.GEMM_NT, enc2d, ww, n * Ls, 512, 512, 512, 512, out=K[li], ldc=512, bias=w.bias.detach())
    print("12 encoder-side key / value projections:", t(kv))
    def caches():
        return torch.zeros((6, 640, 160, 512), dtype=torch.bfloat16, device=dev), torch.zeros((6, 640, 160, 512), dtype=torch.bfloat16, device=dev)
    print("zeroed caches:", t(caches))
