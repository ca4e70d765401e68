"""Style-transfer fine-tuning step (train_decoder.py's compute_epoch, training branch) at the benchmark
shape: Transformer-base 6+6, V = 35 000, 640 sentences of U[28,50] tokens, bf16; classifier with the
reference's defaults (word_vec_size 300, 100 filters of size 5, sequence_length 50).  Prints ms/step and
target tokens/s; with --breakdown also the time of the classifier path alone."""
import argparse, os, sys, time
from argparse import Namespace
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from specialk_amd import train_decoder as TD
from specialk_amd.classifier.CNNModels import ConvNet

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--breakdown", action="store_true")
a = ap.parse_args()
cfg = bench.CONFIGS["cfg2"]
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
m, opt = bench.build(cfg, dt, "cuda")
m.encoder.eval()
torch.manual_seed(7)
cnn = ConvNet(Namespace(num_filters=100, filter_size=5, sequence_length=50, word_vec_size=300, dropout=0.2, num_classes=2),
              cfg["vocab"]).cuda().eval()
cnn.compute_dtype = dt
cnn.refs = Namespace(tgt_label=1)
class_input = torch.nn.Sequential(torch.nn.Linear(cfg["d_model"], cfg["vocab"])).cuda()
trainer = SimpleNamespace(model=m, optimiser=opt, device=torch.device("cuda"), packed=True, tgt_bpe=None,
                          opt=Namespace(label_smoothing=True))
g = torch.Generator().manual_seed(7)
batches = []
for _ in range(4):
    s, sp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
    t, tp = bench.make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
    batches.append(tuple(x.cuda() for x in (s, sp, t, tp)))
ntok = sum(int((b[2][:, 1:] != 0).sum()) for b in batches) / len(batches)


def run(n):
    out = TD.compute_epoch(trainer, class_input, cnn, [batches[i % 4] for i in range(n)])
    return out


run(a.warmup)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = run(a.steps)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / a.steps
print(f"style step: {ms:.2f} ms/step, {ntok / ms * 1e3:,.0f} target tokens/s  (l_recon/word {out[1][-1] / ntok:.3f}, l_class {out[2][-1]:.4f})")
if a.breakdown:
    from specialk_amd.lib.transformer.Models import plan_packing
    b = batches[0]
    tgt_in = b[2][:, :-1].contiguous()
    (plan,) = plan_packing([tgt_in], [0])
    x = torch.randn(plan.n, cfg["d_model"], device="cuda").to(dt)
    for _ in range(3):
        cnn.loss_from_decoder(x, plan.B, plan.L, class_input, 1, off=plan.off)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        cnn.loss_from_decoder(x, plan.B, plan.L, class_input, 1, off=plan.off)
    torch.cuda.synchronize()
    print(f"classifier path alone: {(time.perf_counter() - t0) * 50:.3f} ms on {plan.n} rows")
