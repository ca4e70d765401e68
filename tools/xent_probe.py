"""Times the loss kernel against plain memory traffic of the same size (one GPU).

    python tools/xent_probe.py [rows] [V]
Prints microseconds for: an in-place elementwise pass over the logits (read + write), a read-only
reduction, the loss kernel with and without the gradient, and with a fraction of pad rows.
"""
import sys
import torch
from specialk_amd import _lib as L

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
V = int(sys.argv[2]) if len(sys.argv) > 2 else 35000
dev = torch.device("cuda:0")
x = (torch.randn(rows, V, device=dev) * 2).to(torch.bfloat16)
gold = torch.randint(3, V, (rows,), device=dev)
stats = torch.zeros(3, dtype=torch.float64, device=dev)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


gb = rows * V * 2 / 1e9
t = timed(lambda: x.mul_(1.0))
print(f"in-place mul_      {t:8.1f} us  {2 * gb / t * 1e3:6.2f} TB/s (read+write)")
y = torch.empty_like(x)
t = timed(lambda: y.copy_(x))
print(f"copy               {t:8.1f} us  {2 * gb / t * 1e3:6.2f} TB/s (read+write)")
t = timed(lambda: y.zero_())
print(f"zero fill          {t:8.1f} us  {gb / t * 1e3:6.2f} TB/s (write)")
t = timed(lambda: x.view(torch.int16).amax())
print(f"amax (read only)   {t:8.1f} us  {gb / t * 1e3:6.2f} TB/s (read)")
t = timed(lambda: L.xent_fwd_bwd(x, gold, None, stats, 1.0, 0.1, 0))
print(f"loss, no gradient  {t:8.1f} us  {gb / t * 1e3:6.2f} TB/s (read)")
t = timed(lambda: L.xent_fwd_bwd(x, gold, y, stats, 1.0, 0.1, 0))
print(f"loss + gradient    {t:8.1f} us  {2 * gb / t * 1e3:6.2f} TB/s (read+write)")
x2 = x.clone()
t = timed(lambda: L.xent_fwd_bwd(x2, gold, x2, stats, 1.0, 0.1, 0))
print(f"loss + grad inplace{t:8.1f} us  {2 * gb / t * 1e3:6.2f} TB/s (read+write)")
g2 = gold.clone(); g2[torch.rand(rows, device=dev) < 0.2] = 0
t = timed(lambda: L.xent_fwd_bwd(x, g2, y, stats, 1.0, 0.1, 0))
print(f"loss + grad 20%pad {t:8.1f} us")
