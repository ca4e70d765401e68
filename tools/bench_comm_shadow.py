"""How the benchmark training step reacts to another kernel holding CUs while it runs — what RCCL's all-reduce
kernels do during the backward pass of a data-parallel step — and what leaving those CUs out of the GEMMs' round
planning (skhip_set_reserved_cus) recovers.  A stand-in kernel (tools/occupy.hip: n workgroups spinning for a
stated time) is re-launched back to back on a second, default-priority stream for the whole step, i.e. the worst
case: real collectives hold their CUs for part of the backward pass only.

  hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/occupy.hip -o tools/liboccupy.so
  python tools/bench_comm_shadow.py [blocks ...]          (default 0 8 16 32)
"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import _lib as L, ops

occ = ctypes.CDLL(os.path.join(ROOT, "tools", "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
blocks_list = [int(x) for x in sys.argv[1:]] or [0, 8, 16, 32]
cfg = bench.CONFIGS["cfg2"]
dev = torch.device("cuda", 0)
m, opt = bench.build(cfg, torch.bfloat16, dev)
ops.manual_seed(1000)
g = torch.Generator().manual_seed(7)
batches = []
for _ in range(4):
    s, sp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
    t, tp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
    batches.append(tuple(x.to(dev) for x in (s, sp, t, tp)))
side = torch.cuda.Stream(dev)
sink = torch.zeros(1, dtype=torch.int32, device=dev)
STEPS, SLICE_US = 4, 1000


def run(n_occ, reserve):
    L.set_reserved_cus(reserve)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if n_occ:  # enough back-to-back slices to cover the steps (they keep their CUs until they end)
        for _ in range(int(STEPS * 30e3 / SLICE_US)):
            assert occ.occupy(n_occ, SLICE_US, 24576, sink.data_ptr(), side.cuda_stream) == 0
    e0.record()
    for i in range(STEPS):
        bench.train_step(m, opt, batches[i % 4], None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / STEPS


for _ in range(3):
    run(0, 0)
print(f"{'occupied':>8s} {'reserved':>8s} {'ms/step':>8s}")
for n_occ in blocks_list:
    for reserve in sorted({0, n_occ}):
        ts = sorted(run(n_occ, reserve) for _ in range(5))
        print(f"{n_occ:8d} {reserve:8d} {ts[2]:8.3f}", flush=True)
L.set_reserved_cus(0)
