"""Phase timing inside skhip_decode_stack (SKHIP_STACK_STAMPS=1): runs the worst-case decode shape for a number of
steps and prints, for layer 3 of team 0 / member 0 at the last step, the time between phase boundaries."""
import os, sys
os.environ["SKHIP_STACK_STAMPS"] = "1"
os.environ["SK_DECODE_GRAPH"] = "0"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from specialk_amd import _lib as L
from specialk_amd.decode import BeamDecoder

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 75
cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = 150
m, _ = bench.build(cfg, torch.bfloat16, "cuda")
m.eval()
g = torch.Generator().manual_seed(3)
s, sp = bench.make_batch(g, 128, 12, 42, cfg["vocab"])
dec = BeamDecoder(m, 5, steps)
orig = L.decode_stack
last = {}
def spy(*a, **k):
    last["ws"] = a[12]
    return orig(*a, **k)
L.decode_stack = spy
dec.run(s.cuda(), sp.cuda(), 1)
torch.cuda.synchronize()
st = last["ws"][512:512 + 17 * 8].view(torch.int64).cpu().tolist()
names = ["A qkv", "self-attn", "sync1", "C fc", "sync2", "LN+crossq", "cross-attn", "sync3(+load)", "E fc", "sync4",
         "LN+F w1", "G w2", "sync5", "reduce", "sync6", "LN"]
print(f"history length {steps}; layer 3, team 0, member 0 (10 ns ticks -> us)")
for i in range(16):
    print(f"  {names[i]:14s} {(st[i + 1] - st[i]) / 100:8.2f} us")
print(f"  layer total    {(st[16] - st[0]) / 100:8.2f} us")
