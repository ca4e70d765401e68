#!/usr/bin/env python3
"""Phase times inside the row-panel kernels from in-kernel clock readings (skhip_rowpanel_set_stamps): per wave,
clocks between the phase boundaries, median over workgroups.  Diagnostics only: the readings cost a few scalar
instructions per boundary.

  python tools/rowpanel_stamps.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from specialk_amd import _lib as L, ops  # noqa: E402

T, d, F = 25639, 512, 2048
dev = "cuda"
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
nblk = (T + 103) // 104
x512 = (torch.randn(T, 512, generator=g) * 0.5).to(bf).to(dev)
W1 = torch.nn.Parameter((torch.randn(F, d, generator=g) * d ** -0.5).to(dev))
W2 = torch.nn.Parameter((torch.randn(d, F, generator=g) * F ** -0.5).to(dev))
b1, b2 = torch.zeros(F, device=dev), torch.zeros(d, device=dev)
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
signs = L.rowpanel_pattern(T, F, dev)
f1 = ops.derived_weight("fm", [W1])
out2048 = torch.empty(T, F, dtype=bf, device=dev)
stamps = torch.zeros(nblk * 8 * 32, dtype=torch.int64, device=dev)


def run(name, fn, nslots):
    stamps.zero_()
    for _ in range(40):   # the stamped launch runs behind these, at the clock the chip holds under that load
        fn()
    L.rowpanel_set_stamps(stamps)
    fn()
    L.rowpanel_set_stamps(None)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = stamps.view(nblk, 8, 32).cpu()
    t0 = s[:, :, 0].min()
    life = (s[:, :, 30] - s[:, :, 0]).double()
    ghz = (life / s[:, :, 31].double().clamp_min(1) * 0.1).median()
    print(f"== {name}: first start to last end {int(s[:, :, 30].max() - t0)} clocks, workgroup starts spread over "
          f"{int(s[:, 0, 0].max() - t0)}, wave lifetime median {int(life.median())} max {int(life.max())}, clock {ghz:.2f} GHz")
    blk_life = (s[:, :, 30].max(dim=1).values - s[:, :, 0].min(dim=1).values)
    start = s[:, :, 0].min(dim=1).values - s[:, :, 0].min()
    order = blk_life.argsort(descending=True)
    q = blk_life.double().quantile(torch.tensor([0.0, 0.25, 0.5, 0.75, 0.9, 1.0], dtype=torch.float64)).tolist()
    print("  workgroup lifetimes min/25/50/75/90/max: " + " ".join(f"{int(v)}" for v in q))
    print("  slowest workgroups (id: lifetime, start offset, wave-0 phases): ")
    for b in order[:4].tolist():
        ph = (s[b, 0, 1:nslots] - s[b, 0, 0:nslots - 1]).tolist()
        print(f"    {b}: {int(blk_life[b])} start +{int(start[b])} | " + " ".join(f"{int(v)}" for v in ph))
    print(f"  start offsets min/median/max: {int(start.min())} {int(start.median())} {int(start.max())}; by xcd (id % 8) median lifetime: "
          + " ".join(f"{int(blk_life[i::8].median())}" for i in range(8)))
    for w in (0, 3, 7):
        deltas = (s[:, w, 1:nslots] - s[:, w, 0:nslots - 1])
        med = deltas.median(dim=0).values.tolist()
        print(f"  wave {w}: " + " ".join(f"{int(v):6d}" for v in med))


RB = L.EPI_RELU | L.EPI_RELU_BITS_OUT
run("panel N 2048 relu pattern (start, prologue, [K loop, epilogue] x 4)",
    lambda: L.rowpanel_gemm(x512, f1, F, bias=b1, out=out2048, ldc=F, signs=signs, epilogue=RB), 10)
run("panel N 512 + LayerNorm (start, prologue, K loop, tail)",
    lambda: L.rowpanel_gemm_ln(x512, ops.derived_weight("fm", [torch.nn.Parameter(W1[:512].detach().clone())]), b2, x512, gamma, beta, None, 1e-5, 0.1, 1), 4)
