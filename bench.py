#!/usr/bin/env python3
"""Headline benchmark: Transformer-base training step throughput (train target tokens/sec).

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full optimiser step of the hot path on one synthetic batch per GPU:
zero_grad, encoder-decoder forward, label-smoothed loss, backward, (N>1: bucketed RCCL SUM
all-reduce of gradients overlapped with backward), fused Adam with the Noam schedule, dropout
on as in the reference's training mode.  Workload: 6+6 layers,
d_model 512, 8 heads x 64, FFN 2048, vocab 35000 untied, bf16 compute with fp32 master weights,
640 sentences per GPU with lengths U[28,50] (+SOS/EOS) = ~25k non-pad target tokens, inputs
resident in HBM before the timed region.  Weak scaling: per-GPU batch fixed.

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline:     bf16 MFMA GEMM kernel, algorithmic FLOPs (non-pad rows only) / HIP-event time
  cpu_baseline: the numpy oracle (a port of the reference's algorithm) timed on host cores on
                a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (n_layers, d_model, n_head, d_k, d_inner, vocab, tied, batch, len_lo, len_hi)
    "cfg2": dict(n_layers=6, d_model=512, n_head=8, d_k=64, d_inner=2048, vocab=35000, tied=False, batch=640,
                 lo=28, hi=50, desc="Transformer-base 6+6 d512 FFN2048 V35000, 640 sent/GPU (~25k tgt tokens)"),
    "cfg1": dict(n_layers=2, d_model=256, n_head=8, d_k=32, d_inner=1024, vocab=4096, tied=False, batch=64,
                 lo=6, hi=24, desc="2-layer d256 8-head V4096, 64 sentences (Multi30k-shaped)"),
    "cfg4": dict(n_layers=6, d_model=1024, n_head=16, d_k=64, d_inner=4096, vocab=44544, tied=True, batch=320,
                 lo=28, hi=50, desc="Transformer-big 6+6 d1024 16-head FFN4096 V44544 tied, 320 sent/GPU"),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X
PAD, SOS, EOS = 0, 2, 3


def make_batch(g, n, lo, hi, vocab):
    lens = torch.randint(lo, hi + 1, (n,), generator=g)
    L = int(lens.max()) + 2
    seq = torch.zeros(n, L, dtype=torch.long)
    ids = torch.randint(5, vocab, (n, L), generator=g)
    ar = torch.arange(L)[None, :]
    body = (ar >= 1) & (ar <= lens[:, None])
    seq[body] = ids[body]
    seq[:, 0] = SOS
    seq[torch.arange(n), lens + 1] = EOS
    pos = torch.where(seq != PAD, ar + 1, torch.zeros_like(seq))
    return seq, pos


def build(cfg, dtype, device):
    from specialk_amd.lib.transformer.Models import Transformer
    from specialk_amd.lib.transformer.Optim import ScheduledOptim
    from specialk_amd.optim import FusedAdam
    torch.manual_seed(1234)
    m = Transformer(cfg["vocab"], cfg["vocab"], cfg["hi"] + 3, d_word_vec=cfg["d_model"], d_model=cfg["d_model"],
                    d_inner=cfg["d_inner"], n_layers=cfg["n_layers"], n_head=cfg["n_head"], d_k=cfg["d_k"],
                    d_v=cfg["d_k"], dropout=0.1, tgt_emb_prj_weight_sharing=cfg["tied"],
                    emb_src_tgt_weight_sharing=cfg["tied"])
    for p in m.parameters():  # TransformerModel.initiate (lib/TransformerModel.py:73-75)
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p)
    m = m.to(device).set_compute_dtype(dtype)
    if os.environ.get("SK_BENCH_CHECK_IDS", "1") == "0":  # (the trainer keeps the id / position range check: so does the bench)
        m.encoder.check_ids = m.decoder.check_ids = False
    m.train()
    opt = ScheduledOptim(FusedAdam(filter(lambda x: x.requires_grad, m.parameters()), betas=(0.9, 0.98), eps=1e-09),
                         cfg["d_model"], 4000)
    return m, opt


BATCHED_WGRADS = os.environ.get("SK_BENCH_BATCHED_WGRADS", "1") != "0"
PACKED = True  # run the network on the batch without its padding rows (--padded turns it off)


def train_step(m, opt, batch, reducer, plans=None):
    """One optimiser step.  `plans`: packing plans of this batch started a step earlier (Transformer.plan_ahead);
    without them the step derives them itself and waits for the host copy."""
    from specialk_amd import ops
    src_seq, src_pos, tgt_seq, tgt_pos = batch
    if reducer is not None:
        reducer.zero_grad()
    else:
        opt.zero_grad()
    if PACKED:  # sentence lengths are derived from the ids on the device inside the timed step
        pred, gold = m.forward_packed(src_seq, src_pos, tgt_seq, tgt_pos, plans=plans)
        loss, stats = ops.xent(pred, gold, True, unit_grad=True)
    else:
        pred = m(src_seq, src_pos, tgt_seq, tgt_pos)
        loss, stats = ops.xent(pred.view(-1, pred.size(2)), tgt_seq[:, 1:], True, unit_grad=True)
    if BATCHED_WGRADS:  # as TransformerModel.compute_epoch does
        with ops.batched_weight_grads(on_ready=reducer.mark_ready if reducer is not None else None):
            loss.backward()
    else:
        loss.backward()
    if reducer is not None:
        reducer.finish()
    opt.step_and_update_lr()
    return stats


def gemm_roofline(m, opt, batch, steps):
    """Instrumented extra steps (after the timed region, same process, same data): every bf16
    GEMM launch is bracketed by HIP events on the launch stream.  Algorithmic FLOPs credit only
    non-pad rows: padding work earns no credit."""
    from specialk_amd import _lib as L
    src_seq, _, tgt_seq, _ = batch
    Ts, Tt = src_seq.numel(), tgt_seq[:, :-1].numel()
    frac = {Ts: float((src_seq != PAD).sum()) / Ts}
    ft = float((tgt_seq[:, :-1] != PAD).sum()) / Tt
    frac[Tt] = ft if Tt not in frac else 0.5 * (ft + frac[Tt])
    records = []
    orig = L.gemm

    def timed(layout, A, B, M, N, K, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(layout, A, B, M, N, K, *a, **kw)
        e1.record()
        f = frac.get(M, frac.get(K, 1.0))
        records.append((e0, e1, 2.0 * M * N * K * f, A.dtype))
        return out

    orig_grouped = L.gemm_grouped_tn

    def timed_grouped(problems):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_grouped(problems)
        e1.record()
        fl = sum(2.0 * M * N * K * frac.get(K, 1.0) for _, _, _, M, N, K, *_ in problems)
        records.append((e0, e1, fl, problems[0][0].dtype))

    orig_panel, orig_panel_ln = L.rowpanel_gemm, L.rowpanel_gemm_ln

    def timed_panel(a, w_fm, N, *args, **kw):  # (T, 512) x (N, 512)^T
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_panel(a, w_fm, N, *args, **kw)
        e1.record()
        records.append((e0, e1, 2.0 * a.shape[0] * N * 512 * frac.get(a.shape[0], 1.0), a.dtype))
        return out

    def timed_panel_ln(a, *args, **kw):  # the output projection with its LayerNorm tail: credited as the product alone
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_panel_ln(a, *args, **kw)
        e1.record()
        records.append((e0, e1, 2.0 * a.shape[0] * 512 * 512 * frac.get(a.shape[0], 1.0), a.dtype))
        return out

    L.gemm, L.gemm_grouped_tn, L.rowpanel_gemm, L.rowpanel_gemm_ln = timed, timed_grouped, timed_panel, timed_panel_ln
    try:
        for _ in range(steps):
            train_step(m, opt, batch, None)
        torch.cuda.synchronize()
    finally:
        L.gemm, L.gemm_grouped_tn, L.rowpanel_gemm, L.rowpanel_gemm_ln = orig, orig_grouped, orig_panel, orig_panel_ln
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in records)
    flops = sum(f for _, _, f, _ in records)
    n = len(records)
    # HBM-side bytes per GEMM launch from the committed counter passes of this same command (rocprofv3 --pmc
    # FETCH_SIZE and --pmc WRITE_SIZE, separate runs; tools/pmc_hbm_traffic.py): counters cannot be read in-process
    traffic, traffic_src = None, None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc_hbm_traffic.json")
    if os.path.exists(pmc):
        with open(pmc) as f:
            traffic = round(json.load(f)["hbm_bytes_per_launch"])
        traffic_src = "profiles/r03_pmc_hbm_traffic.json (bytes per launch, cfg2)"
    out = _roofline_dict(flops, ms, n, steps)
    out["traffic"], out["traffic_source"] = traffic, traffic_src
    return out


def _roofline_dict(flops, ms, n, steps):
    return dict(bound="mfma", achieved=round(flops / (ms * 1e-3) / 1e12, 2), peak=MFMA_BF16_PEAK_TFLOPS,
                unit="TFLOP/s", frac=round(flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None,
                kernel="b16x/b16g/b16::gemm_kernel + b16x::gemm_grouped_tn_kernel + rp::panel_kernel (bf16 MFMA GEMMs, all layouts; the panel kernel's LayerNorm tail is inside its time)", launches_per_step=n // steps,
                avg_launch_us=round(ms * 1e3 / n, 2), avg_algorithmic_gflop_per_launch=round(flops / n / 1e9, 3))


def cpu_baseline(cfg, seconds_budget=25.0):
    """The numpy oracle (port of the reference algorithm) on host cores: full training steps
    (forward, loss, backward, Adam) of the SAME architecture on a small batch of the same length
    distribution, repeated until ~seconds_budget."""
    from oracle import transformer_np as O
    import contextlib
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    cores = min(16, os.cpu_count() or 1)
    rng = np.random.default_rng(0)
    V, d, F, NL, H, dk = cfg["vocab"], cfg["d_model"], cfg["d_inner"], cfg["n_layers"], cfg["n_head"], cfg["d_k"]
    def mat(*shape):
        return (rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] + shape[1]))).astype(np.float32)
    w = {"encoder.src_word_emb.weight": mat(V, d), "decoder.tgt_word_emb.weight": mat(V, d), "generator.weight": mat(V, d),
         "encoder.position_enc.weight": O.sinusoid_table(cfg["hi"] + 4, d, 0),
         "decoder.position_enc.weight": O.sinusoid_table(cfg["hi"] + 4, d, 0)}
    for side, kinds in (("encoder", ("slf_attn",)), ("decoder", ("slf_attn", "enc_attn"))):
        for i in range(NL):
            pre = f"{side}.layer_stack.{i}."
            for a in kinds:
                for nm in ("w_qs", "w_ks", "w_vs"):
                    w[pre + a + f".{nm}.weight"] = mat(H * dk, d); w[pre + a + f".{nm}.bias"] = np.zeros(H * dk, np.float32)
                w[pre + a + ".fc.weight"] = mat(d, H * dk); w[pre + a + ".fc.bias"] = np.zeros(d, np.float32)
                w[pre + a + ".layer_norm.weight"] = np.ones(d, np.float32); w[pre + a + ".layer_norm.bias"] = np.zeros(d, np.float32)
            w[pre + "pos_ffn.w_1.weight"] = mat(F, d)[:, :, None]; w[pre + "pos_ffn.w_1.bias"] = np.zeros(F, np.float32)
            w[pre + "pos_ffn.w_2.weight"] = mat(d, F)[:, :, None]; w[pre + "pos_ffn.w_2.bias"] = np.zeros(d, np.float32)
            w[pre + "pos_ffn.layer_norm.weight"] = np.ones(d, np.float32); w[pre + "pos_ffn.layer_norm.bias"] = np.zeros(d, np.float32)
    ocfg = O.Config(NL, H, dk, dk, d, F)
    g = torch.Generator().manual_seed(7)
    n_sent = 64
    src_seq, src_pos = (x.numpy() for x in make_batch(g, n_sent, cfg["lo"], cfg["hi"], V))
    tgt_seq, tgt_pos = (x.numpy() for x in make_batch(g, n_sent, cfg["lo"], cfg["hi"], V))
    trainable = [k for k in w if "position_enc" not in k]
    mom = {k: np.zeros_like(w[k]) for k in trainable}
    var = {k: np.zeros_like(w[k]) for k in trainable}

    def step(i):
        logits, cache = O.transformer_fwd(w, ocfg, src_seq, src_pos, tgt_seq, tgt_pos)
        _, dpred, _ = O.ls_xent(logits.reshape(-1, V), tgt_seq[:, 1:].reshape(-1), True)
        grads = O.transformer_bwd(w, ocfg, dpred.reshape(logits.shape), cache)
        lr = np.float32(O.noam_lr(i, d, 4000))
        for k in trainable:
            O.adam_step(w[k], grads[k].astype(np.float32), mom[k], var[k], i, lr)

    with (threadpool_limits(limits=cores) if threadpool_limits else contextlib.nullcontext()):
        step(1)  # warm-up
        t0, n = time.time(), 0
        while n < 2 or (time.time() - t0 < seconds_budget and n < 20):
            step(n + 2)
            n += 1
        dt = time.time() - t0
    tok = int((tgt_seq[:, 1:] != PAD).sum())
    return dict(value=round(tok * n / dt, 1), unit="target tokens/sec", cores=cores, kind="port",
                sample=f"{n} full training steps (fwd+loss+bwd+Adam, fp32 numpy) of the same architecture on "
                       f"{n_sent} sentences/step ({tok} target tokens), lengths U[{cfg['lo']},{cfg['hi']}], {dt:.1f} s")


def decode_leg(device, n_batches=10, batch=128, beam=5, max_len=150, copy_task_steps=600):
    """Beam-search decode throughput at the style-transfer inference shape (beam 5, 128 sentences per batch, source
    lengths U[12,42], at most 150 target tokens; lib/transformer/Translator.py:55-226), inputs resident in HBM.

    "worst_case": the benchmark architecture (6+6, d 512) with random weights — such a network never emits EOS, so
    every sentence decodes all 150 steps with all 640 beams alive.  "copy_task": a 2+2-layer network of the same
    width, trained here (untimed) with this package's own training step on a synthetic copy task until it
    reproduces its input, so that hypotheses end with EOS at about the source length (the 6+6 post-LN stack does
    not leave that task's initial plateau within a few thousand steps, in fp32 or bf16).
    HBM roofline of the worst case: algorithmic bytes of one step at history length t — self-attention keys and
    values of t positions for every beam row, encoder-side keys / values once per sentence, the decoder's and
    generator's bf16 weights once, the fp32 log-probabilities written once (the top-k reads chunk statistics and a few
    chunks per row) —
    summed over the steps, divided by the time."""
    from specialk_amd import ops
    from specialk_amd.decode import BeamDecoder
    out = {}
    g = torch.Generator().manual_seed(3)

    def run(m, vocab_used, n_batches):
        dec = BeamDecoder(m, beam, max_len)
        sent = tok = exact = 0
        steps_bytes = 0.0
        dt = 0.0
        for i in range(n_batches + 1):  # batch 0 warms up
            s, sp = make_batch(g, batch, 12, 42, vocab_used)
            s, sp = s.to(device), sp.to(device)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hyps, _ = dec.run(s, sp, 1)
            torch.cuda.synchronize()
            if i == 0:
                continue
            dt += time.perf_counter() - t0
            sent += len(hyps)
            tok += sum(len(h[0]) for h in hyps)
            src = s.cpu().tolist()
            exact += sum(1 for row, h in zip(src, hyps) if [t for t in row[1:] if t != 0] == list(h[0]))
            steps_bytes += decode_step_bytes(m, batch, beam, int((s != PAD).sum(1).max()), dec.last_steps)
        return sent, tok, exact, dt, steps_bytes

    cfg = dict(CONFIGS["cfg2"]); cfg["hi"] = max_len
    m, _ = build(cfg, torch.bfloat16, device)
    m.eval()
    sent, tok, exact, dt, nbytes = run(m, cfg["vocab"], n_batches)
    traffic, traffic_src = None, None  # counter passes of this same leg, committed (counters cannot be read in-process)
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc_decode_hbm_traffic.json")
    if os.path.exists(pmc) and (batch, beam, max_len) == (128, 5, 150):
        with open(pmc) as f:
            traffic = round(json.load(f)["hbm_bytes_per_step"])
        traffic_src = "profiles/r03_pmc_decode_hbm_traffic.json (bytes per step)"
    out["worst_case"] = dict(model="6+6 d512 FFN2048 V35000, random weights (no EOS: all 150 steps, 640 beams alive)",
                             sentences=sent, sentences_per_sec=round(sent / dt, 1), tokens_per_sec=round(tok / dt, 1),
                             mean_output_length=round(tok / sent, 1), ms_per_step=round(dt / (n_batches * max_len) * 1e3, 4),
                             roofline=dict(bound="hbm", achieved=round(nbytes / dt / 1e9, 1), peak=8000.0, unit="GB/s",
                                           frac=round(nbytes / dt / 8e12, 4), traffic=traffic, traffic_source=traffic_src,
                                           algorithmic_bytes_per_step=round(nbytes / (n_batches * max_len))))
    del m
    if copy_task_steps:
        cfg2 = dict(cfg); cfg2["n_layers"] = 2
        m, opt = build(cfg2, torch.bfloat16, device)
        opt.n_warmup_steps = 1500
        ops.manual_seed(5)
        # at least copy_task_steps steps, then on until 100 consecutive steps predicted every token (the step at
        # which the network leaves the task's initial plateau varies from run to run: the embedding gradient is
        # summed with atomics), at most 5 x copy_task_steps
        trained, clean, recent = 0, 0, []
        while trained < 5 * copy_task_steps and (trained < copy_task_steps or clean < 100):
            s_, sp_ = make_batch(g, 640, 12, 42, 2000)
            recent.append(train_step(m, opt, tuple(x.to(device) for x in (s_, sp_, s_, sp_)), None))
            trained += 1
            if len(recent) == 20:  # one host read per 20 steps
                for _, n_correct, n_word in torch.stack(recent).tolist():
                    clean = clean + 1 if n_correct == n_word else 0
                recent = []
        m.eval()
        sent, tok, exact, dt, _ = run(m, 2000, n_batches)
        out["copy_task"] = dict(model=f"2+2 d512 FFN2048 V35000 trained {trained} steps on a synthetic copy task",
                                sentences=sent, sentences_per_sec=round(sent / dt, 1), tokens_per_sec=round(tok / dt, 1),
                                mean_output_length=round(tok / sent, 1), exact_copies=exact)
    out["config"] = dict(beam=beam, batch=batch, src_len="U[12,42]", max_token_seq_len=max_len, dtype="bf16")
    return out


def decode_step_bytes(m, n, bm, Ls, steps):
    """Algorithmic HBM bytes of `steps` decode steps of a batch (see decode_leg)."""
    dec = m.decoder
    NL = len(dec.layer_stack)
    a = dec.layer_stack[0].slf_attn
    H, dk, dv = a.n_head, a.d_k, a.d_v
    d = dec.tgt_word_emb.weight.shape[1]
    F = dec.layer_stack[0].pos_ffn.w_1.weight.shape[0]
    V = m.generator.weight.shape[0]
    R = n * bm
    kv_row = H * (dk + dv) * 2                                   # bytes of one cached position (bf16)
    # per layer: self-attention q, k, v, output projections; cross-attention q and output projections; the two FFN matrices
    weights = NL * (3 * H * dk * d + 3 * H * dv * d + 2 * d * F) * 2 + V * d * 2
    # fp32 log-probabilities written once; the top-k reads their 64-column chunk statistics and a few chunks per row
    per_step_fixed = weights + NL * n * Ls * kv_row + R * V * 4 + 2 * R * ((V + 63) // 64) * 8 + R * 8 * 64 * 4
    hist = NL * R * kv_row * (steps * (steps + 1) / 2)           # sum over t = 1..steps of t positions
    return steps * per_step_fixed + hist


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N ranks (one per GPU) as children of
    torch.distributed.run with this script's own arguments, relay rank 0's result line and return the exit code to
    use.  The launcher process makes no GPU call (importing torch does not initialise the device)."""
    import socket
    import subprocess
    if torch.cuda.device_count() < n and "SK_BENCH_DEVICE" not in os.environ:
        print(f"bench.py: --gpus {n} but {torch.cuda.device_count()} visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        print(f"bench.py: the {n}-rank run failed (exit {proc.returncode}, {len(lines)} result lines)", file=sys.stderr)
        return proc.returncode or 1
    out = json.loads(lines[-1])
    if out.get("n_gpus") != n or out.get("ranks_seen") != n:
        print(f"bench.py: asked for {n} ranks, the result reports n_gpus={out.get('n_gpus')} "
              f"ranks_seen={out.get('ranks_seen')}", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--padded", action="store_true", help="keep padding rows (the reference's tensor shapes)")
    ap.add_argument("--no-decode", action="store_true", help="skip the beam-decode leg")
    ap.add_argument("--decode-only", action="store_true",
                    help="profiling aid: run only the beam-decode leg (worst case, 3 batches) and print its result")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: this process becomes the launcher (it never touches a
        # GPU) and the N ranks are fresh children of torch.distributed.run, as lib/train.py:385-388's fan-out over
        # the visible devices would be; rank 0's one result line is relayed, any failing rank fails the run
        raise SystemExit(launch_ranks(args.gpus))

    # stdout carries exactly one line, the result: libraries that write to file descriptor 1 themselves (RCCL
    # prints a version banner at start-up) are pointed at stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("SK_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))  # override: rehearsals on one card
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # SK_BENCH_FORCE_DIST=1: run the N > 1 flow (process group, bucketed all-reduce, barriers, reductions of the
    # results) in a group of ONE rank — a rehearsal of the RCCL calls on a single card
    dist_on = world > 1 or os.environ.get("SK_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("SK_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 flow on one card
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": device} if backend == "nccl" else {}))

    if args.decode_only:
        out = decode_leg(device, n_batches=3, copy_task_steps=0)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        return
    cfg = CONFIGS[args.config]
    global PACKED
    PACKED = not args.padded
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    m, opt = build(cfg, dtype, device)
    from specialk_amd import ops
    from specialk_amd.parallel import GradBucketReducer
    ops.manual_seed(1000 + rank)
    reducer = GradBucketReducer(list(m.parameters()), always_reduce=True) if dist_on else None

    g = torch.Generator().manual_seed(7 + 1000 * rank)
    batches = []
    for _ in range(4):
        s, sp = make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
        t, tp = make_batch(g, cfg["batch"], cfg["lo"], cfg["hi"], cfg["vocab"])
        batches.append(tuple(x.to(device) for x in (s, sp, t, tp)))
    tok_per_batch = [int((b[2][:, 1:] != PAD).sum()) for b in batches]

    # Every step derives the packing plan (sentence lengths, row offsets; two small kernels and one host read) of the
    # NEXT step's batch before it runs its own, as a loader prefetches: the host read then never stalls the queue.
    def step(i, pending):
        nxt = m.plan_ahead(*batches[(i + 1) % len(batches)]) if PACKED else None
        train_step(m, opt, batches[i % len(batches)], reducer, plans=pending)
        return nxt

    pending = m.plan_ahead(*batches[0]) if PACKED else None
    for i in range(args.warmup):
        pending = step(i, pending)
    from specialk_amd import ops as _ops
    _ops.settle_host_garbage()  # as the trainer does at the start of an epoch: no full garbage collection (100 ms) inside a step
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    for i in range(args.warmup, args.warmup + args.steps):
        pending = step(i, pending)
        tokens += tok_per_batch[i % len(batches)]
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    ranks_seen, rank_ms = 1, [round(elapsed / args.steps * 1e3, 3)]
    if dist_on:
        red_dev = device if dist.get_backend() == "nccl" else "cpu"
        ones = torch.ones(1, dtype=torch.float64, device=red_dev)  # how many ranks took part, counted by the collective itself
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
        per_rank = torch.zeros(world, dtype=torch.float64, device=red_dev)
        per_rank[rank] = elapsed / args.steps * 1e3
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        rank_ms = [round(v, 3) for v in per_rank.tolist()]
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ttok = torch.tensor([tokens], dtype=torch.float64, device=red_dev)
        dist.all_reduce(ttok, op=dist.ReduceOp.SUM)
        tokens = float(ttok.item())

    out = None
    if rank == 0:
        out = {
            "metric": "train target tokens/sec", "value": round(tokens / elapsed, 1), "unit": "tokens/sec",
            "n_gpus": world, "ranks_seen": ranks_seen, "ms_per_step_by_rank": rank_ms, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": cfg["desc"], "config_name": args.config, "per_gpu_sentences": cfg["batch"],
                       "target_tokens_per_step": int(tokens / args.steps), "padding_rows": "skipped (packed)" if PACKED else "computed", "label_smoothing": True, "dropout": 0.1,
                       "optimizer": "Adam(0.9,0.98,1e-9)+Noam(4000)", "parallelism": f"dp{world}"},
        }
    if dist_on:
        dist.barrier()
    if rank == 0:
        # rank 0 alone, after the timed region: local steps without the gradient exchange (the other ranks wait
        # at the barrier below), so the figure is the per-GPU kernel efficiency at every N
        if not args.no_roofline and args.dtype == "bf16":
            out["roofline"] = gemm_roofline(m, opt, batches[0], 3)
            if args.config != "cfg2" or args.padded:  # the counter passes were taken on the default workload
                out["roofline"]["traffic"] = out["roofline"]["traffic_source"] = None
        if not args.no_decode and world == 1 and args.config == "cfg2" and args.dtype == "bf16":
            del m, opt
            torch.cuda.empty_cache()
            out["decode"] = decode_leg(device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg)
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
