"""Host-side phase times of BeamDecoder.run on a batch that finishes early (the copy-task shape: about 45 steps):
wall time of the whole call against the time inside the step loop, measured by wrapping the pieces."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import _lib as L
from specialk_amd.decode import BeamDecoder

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 45
cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = 150; cfg["n_layers"] = 2
m, _ = bench.build(cfg, torch.bfloat16, "cuda")
m.eval()
g = torch.Generator().manual_seed(3)
dec = BeamDecoder(m, 5, steps)   # random weights never emit EOS: the run takes exactly `steps` steps
batches = [tuple(x.cuda() for x in bench.make_batch(g, 128, 12, 42, 2000)) for _ in range(6)]
marks = {}
orig_capture = BeamDecoder._capture_pair.__func__


def timed_capture(cls, step, cur, dev):
    torch.cuda.synchronize()
    t = time.perf_counter()
    gr = orig_capture(cls, step, cur, dev)
    torch.cuda.synchronize()
    marks["capture"] = marks.get("capture", 0.0) + time.perf_counter() - t
    return gr


BeamDecoder._capture_pair = classmethod(timed_capture)
for i, (s, sp) in enumerate(batches):
    if i == 1:
        marks.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    dec.run(s, sp, 1)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 5
print(f"{steps} steps per batch: {total * 1e3:.2f} ms per batch, of which graph capture {marks['capture'] / 5 * 1e3:.2f} ms")
os.environ["SK_DECODE_GRAPH"] = "0"
dec2 = BeamDecoder(m, 5, steps)
dec2.run(*batches[0], 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s, sp in batches[1:]:
    dec2.run(s, sp, 1)
torch.cuda.synchronize()
print(f"without the graph: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per batch")
