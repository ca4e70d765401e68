"""Per-shape table of the GEMM launches of one benchmark training step: every skhip_gemm / skhip_rowpanel_gemm[_ln] call is
bracketed by events on the launch stream; rows are grouped by (layout, M, N, K, epilogue).
Usage: python tools/gemm_shapes.py [cfg2] [steps]"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import _lib as L, ops

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
m, opt = bench.build(cfg, torch.bfloat16, dev)
ops.manual_seed(1000)
g = torch.Generator().manual_seed(7)
s, sp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
t, tp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
batch = tuple(x.to(dev) for x in (s, sp, t, tp))
for _ in range(4):
    bench.train_step(m, opt, batch, None)
torch.cuda.synchronize()
recs = []
orig = L.gemm
def timed(layout, A, B, M, N, K, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(layout, A, B, M, N, K, *a, **kw); e1.record()
    recs.append(((layout, M, N, K, kw.get("epilogue", 0), bool(kw.get("split_k", False))), e0, e1))
    return out
orig_g = L.gemm_grouped_tn
def timed_g(problems):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_g(problems); e1.record()
    fl = sum(2.0 * M * N * K for _, _, _, M, N, K, *_ in problems)
    recs.append((("grp", len(problems), int(fl / 1e9), 0, 0, False), e0, e1))
orig_p, orig_pl = L.rowpanel_gemm, L.rowpanel_gemm_ln
def timed_p(a, w_fm, N, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig_p(a, w_fm, N, *args, **kw); e1.record()
    recs.append((("panel", a.shape[0], N, 512, kw.get("epilogue", 0), False), e0, e1))
    return out
def timed_pl(a, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig_pl(a, *args, **kw); e1.record()
    recs.append((("pan+LN", a.shape[0], 512, 512, 0, False), e0, e1))
    return out
L.gemm, L.gemm_grouped_tn, L.rowpanel_gemm, L.rowpanel_gemm_ln = timed, timed_g, timed_p, timed_pl
for _ in range(steps):
    bench.train_step(m, opt, batch, None)
torch.cuda.synchronize()
L.gemm = orig
agg = collections.OrderedDict()
for key, e0, e1 in recs:
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values()) / steps
print(f"GEMM total {tot:.3f} ms/step, {len(recs) // steps} launches/step")
print(f"{'layout':6s} {'M':>7s} {'N':>7s} {'K':>7s} ep sk {'n/step':>6s} {'us':>8s} {'ms/step':>8s} {'TF/s':>7s}")
for (lay, M, N, K, ep, sk), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = ms / n * 1e3
    tf = (N * 1e9 if lay == "grp" else 2.0 * M * N * K) / us / 1e6  # grouped rows: M = problems, N = GFLOP
    print(f"{str(lay):6s} {M:7d} {N:7d} {K:7d} {ep:2d} {int(sk):2d} {n // steps:6d} {us:8.1f} {ms / steps:8.3f} {tf:7.0f}")
