"""cProfile of the inference encoder pass that opens a beam search (cfg-2 model, 128 sentences)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = 150
m, _ = bench.build(cfg, torch.bfloat16, "cuda")
m.eval()
g = torch.Generator().manual_seed(3)
s, sp = bench.make_batch(g, 128, 12, 42, cfg["vocab"])
s, sp = s.cuda(), sp.cuda()
with torch.no_grad():
    for _ in range(5):
        m.encoder(s, sp)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        m.encoder(s, sp)
        torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(30)
