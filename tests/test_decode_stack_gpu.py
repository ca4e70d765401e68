"""The one-launch decoder stack of a beam-search step (skhip_decode_stack) against the op-by-op step it replaces
(projection GEMMs, skhip_decode_attn, skhip_add_ln_fwd — themselves checked against the oracle in
test_kernels_gpu.py / test_model_gpu.py): same model, same batch, the search's choices of the op-by-op run forced
onto the one-launch run so that both see identical histories at every step."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(n_layers, vocab=1000, len_max=40):
    from specialk_amd.lib.transformer.Models import Transformer
    torch.manual_seed(5)
    m = Transformer(vocab, vocab, len_max, d_word_vec=512, d_model=512, d_inner=2048, n_layers=n_layers, n_head=8, d_k=64,
                    d_v=64, tgt_emb_prj_weight_sharing=False, emb_src_tgt_weight_sharing=False)
    for p in m.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p)
        else:  # biases and LayerNorm vectors away from their 0 / 1 defaults
            p.data.add_(0.1 * torch.randn_like(p))
    return m.to(DEV).eval().set_compute_dtype(torch.bfloat16)


def batch(n, lo, hi, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (n,), generator=g)
    L = int(lens.max()) + 2
    seq = torch.zeros(n, L, dtype=torch.long)
    for i, ln in enumerate(lens.tolist()):
        seq[i, 0] = 2
        seq[i, 1:ln + 1] = torch.randint(5, vocab, (ln,), generator=g)
        seq[i, ln + 1] = 3
    pos = (torch.arange(L)[None, :] + 1) * (seq != 0)
    return seq.to(DEV), pos.to(DEV)


@pytest.mark.parametrize("n_sent,n_layers", [(12, 2), (128, 1), (7, 3), (1, 2), (141, 1)])
def test_stack_step_matches_op_by_op(n_sent, n_layers):
    """n_sent x 5 beam rows: 60, 640 (32 teams of 20 rows: waves 0..3 with a third row), 35, 5 (one sentence),
    705 (teams of 23 and 22 rows).  Every step's
    decoder output within 2e-2 of the op-by-op path's (bf16 storage, different summation order in the second
    feed-forward product), over 12 steps of growing history with re-parented beams."""
    from specialk_amd.decode import BeamDecoder
    m = build(n_layers)
    src, pos = batch(n_sent, 3, 17, 1000, n_sent)
    ref = BeamDecoder(m, 5, 12, use_graph=False, use_stack=False)
    ref.trace = []
    hyp_ref, _ = ref.run(src, pos, 2)
    assert not ref.used_stack and len(ref.trace) >= 12
    got = BeamDecoder(m, 5, 12, use_graph=False, use_stack=True)
    got.trace, got.forced = [], [t[1:] for t in ref.trace]
    hyp_got, _ = got.run(src, pos, 2)
    assert got.used_stack
    assert hyp_got == hyp_ref
    worst = 0.0
    for step, (a, b) in enumerate(zip(got.trace, ref.trace)):
        xa, xb = a[0].float(), b[0].float()
        err = float((xa - xb).abs().max() / xb.abs().max())
        worst = max(worst, err)
        assert err <= 2e-2, (step, err)
    print(f"worst relative difference of a step's decoder output: {worst:.4f}")


def test_stack_search_agrees_with_op_by_op_search():
    """Unforced: the two searches pick the same hypotheses for (nearly) every sentence — random weights leave
    near-ties that bf16 rounding order may break differently — and the one-launch path replayed from a HIP graph
    gives the same bits as when launched step by step."""
    from specialk_amd.decode import BeamDecoder
    m = build(2)
    src, pos = batch(40, 3, 17, 1000, 1)
    h0, _ = BeamDecoder(m, 5, 14, use_graph=False, use_stack=False).run(src, pos, 1)
    h1, s1 = BeamDecoder(m, 5, 14, use_graph=False, use_stack=True).run(src, pos, 1)
    h2, s2 = BeamDecoder(m, 5, 14, use_graph=True, use_stack=True).run(src, pos, 1)
    same = sum(a == b for a, b in zip(h0, h1))
    assert same >= 36, same
    assert h1 == h2 and [x.tolist() for x in s1] == [x.tolist() for x in s2]


def test_stack_encoder_attention_forms_agree(monkeypatch):
    """The encoder attention of the one-launch stack has two forms — per (sentence, 32-position segment) with the
    beams of a sentence sharing the loaded keys / values (source sentences of at most 160 positions), and row by row
    (longer ones; SKHIP_STACK_XSHARE=0 forces it): same hypotheses and scores to bf16 rounding on a batch whose
    sentences straddle team boundaries (7 sentences x 5 beams = 35 rows in teams of 18)."""
    from specialk_amd.decode import BeamDecoder
    m = build(2)
    src, pos = batch(7, 3, 17, 1000, 4)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SKHIP_STACK_XSHARE", mode)
        dec = BeamDecoder(m, 5, 12, use_graph=False, use_stack=True)
        dec.trace = []
        hyps, scores = dec.run(src, pos, 2)
        assert dec.used_stack
        res[mode] = (hyps, [x.tolist() for x in scores], [t[0].float() for t in dec.trace])
    same = sum(a == b for a, b in zip(res["1"][0], res["0"][0]))
    assert same >= 6, same   # (random weights leave near-ties that a different summation order may break differently)
    a, b = res["1"][2][0], res["0"][2][0]   # the first step's decoder output: identical histories by construction
    assert float((a - b).abs().max() / b.abs().max()) <= 2e-2


@pytest.mark.parametrize("switch", ["SKHIP_STACK_PIPE", "SKHIP_STACK_FUSERED"])
def test_stack_fallback_forms_match_op_by_op(monkeypatch, switch):
    """The stack keeps two older forms behind switches: self-attention in 64-position sweeps by attend() (what
    history rows too long for the LDS table take; SKHIP_STACK_PIPE=0) and the feed-forward partial products summed by
    columns with a sixth team barrier (SKHIP_STACK_FUSERED=0).  Each against the op-by-op path over 40 steps, the
    search's choices forced to be the same: every step's decoder output within 2e-2."""
    from specialk_amd.decode import BeamDecoder
    m = build(2, len_max=48)
    src, pos = batch(9, 5, 20, 1000, 21)
    ref = BeamDecoder(m, 5, 40, use_graph=False, use_stack=False)
    ref.trace = []
    ref.run(src, pos, 1)
    monkeypatch.setenv(switch, "0")
    got = BeamDecoder(m, 5, 40, use_graph=False, use_stack=True)
    got.trace, got.forced = [], [t[1:] for t in ref.trace]
    got.run(src, pos, 1)
    assert got.used_stack and len(got.trace) == len(ref.trace) >= 30
    for a, b in zip(got.trace, ref.trace):
        xa, xb = a[0].float(), b[0].float()
        assert float((xa - xb).abs().max() / xb.abs().max()) <= 2e-2


def test_stack_long_histories_match_op_by_op():
    """Histories of up to 150 positions (five 32-position stages of the self-attention pipeline, the length of the
    benchmark's decode leg) and source sentences of up to 70 positions (three segments of the encoder attention): every step's
    decoder output of the one-launch stack within 2e-2 of the op-by-op path's, the search's choices forced to be the
    same."""
    from specialk_amd.decode import BeamDecoder
    m = build(1, len_max=160)
    src, pos = batch(6, 30, 70, 1000, 9)
    ref = BeamDecoder(m, 5, 150, use_graph=False, use_stack=False)
    ref.trace = []
    ref.run(src, pos, 1)
    got = BeamDecoder(m, 5, 150, use_graph=False, use_stack=True)
    got.trace, got.forced = [], [t[1:] for t in ref.trace]
    got.run(src, pos, 1)
    assert got.used_stack and len(got.trace) == len(ref.trace) >= 100
    worst = 0.0
    for step, (a, b) in enumerate(zip(got.trace, ref.trace)):
        xa, xb = a[0].float(), b[0].float()
        err = float((xa - xb).abs().max() / xb.abs().max())
        worst = max(worst, err)
        assert err <= 2e-2, (step, err)
    print(f"worst relative difference over {len(ref.trace)} steps: {worst:.4f}")


def test_kept_weights_follow_parameter_updates():
    """A decoder object keeps its prepared weights (stacked q | k | v, fragment-major matrices) between calls: the
    second call on unchanged weights gives the first call's bits, and after the weights change — through torch (version
    counters move) or through this package's optimiser (its kernel moves no version counter; ops.weights_epoch
    does) — a call gives the bits of a fresh decoder object."""
    from specialk_amd import ops
    from specialk_amd.decode import BeamDecoder
    from specialk_amd.optim import FusedAdam
    m = build(2)
    src, pos = batch(6, 3, 17, 1000, 11)

    def first_output(dec):
        dec.trace = []
        dec.run(src, pos, 1)
        assert dec.used_stack
        return dec.trace[0][0].clone()

    dec = BeamDecoder(m, 5, 6, use_graph=False, use_stack=True)
    x0 = first_output(dec)
    held = dec._prep[1]
    assert torch.equal(first_output(dec), x0) and dec._prep[1] is held   # kept, and the same bits
    with torch.no_grad():
        for p in m.decoder.parameters():
            if p.dim() > 1:
                p.mul_(1.05)
    ops.refresh_derived()   # the bf16 working copies follow the parameters
    x1 = first_output(dec)
    assert dec._prep[1] is not held and not torch.equal(x1, x0)
    assert torch.equal(x1, first_output(BeamDecoder(m, 5, 6, use_graph=False, use_stack=True)))
    opt = FusedAdam([p for p in m.parameters()], lr=0.05)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    versions = [p._version for p in m.decoder.parameters()]
    opt.step()
    assert versions == [p._version for p in m.decoder.parameters()]   # the case the update counter exists for
    x2 = first_output(dec)
    assert not torch.equal(x2, x1)
    assert torch.equal(x2, first_output(BeamDecoder(m, 5, 6, use_graph=False, use_stack=True)))



def test_stack_step_matches_bf16_oracle():
    """The one-launch stack against the ORACLE (not against this package's other decode path): at every step of a
    beam search over 12 sentences x 5 beams, the decoder output of each live beam row equals, within the bf16 bar of
    2e-2, the oracle's decoder forward pass over that row's whole prefix (Translator.py:113-138 re-runs the decoder on
    the prefix every step; Models.py:167-199) in its bf16-storage mode, last position; and where the oracle's best
    continuation of a sentence leads its second best by a margin, the search picked exactly that (beam, token)."""
    import numpy as np
    from oracle import transformer_np as O
    from specialk_amd.decode import BeamDecoder
    n, bm, n_layers, V = 12, 5, 2, 1000
    m = build(n_layers, vocab=V)
    src, pos = batch(n, 3, 17, V, 33)
    dec = BeamDecoder(m, bm, 10, use_graph=False, use_stack=True)
    dec.trace = []
    dec.run(src, pos, 1)
    assert dec.used_stack and len(dec.trace) == 10
    w = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
    cfg = O.Config(n_layers, 8, 64, 64, 512, 2048)
    src_np, pos_np = src.cpu().numpy(), pos.cpu().numpy()
    with O.bf16_storage():
        enc = O.encoder_fwd(w, cfg, src_np, pos_np)[0]
    gen = O.round_bf16(w["generator.weight"]).astype(np.float64)
    prefixes = [[[O.SOS]] + [None] * (bm - 1) for _ in range(n)]   # step 1 expands beam 0 only (Beam.py:29-32, 51-54)
    live = [True] * n
    scores_prev = np.zeros((n, bm))
    worst, checked = 0.0, 0
    for step, (x, new_scores, prev, new_tok) in enumerate(dec.trace):
        rows = [(s, k) for s in range(n) if live[s] for k in range(bm) if prefixes[s][k] is not None]
        t = step + 1
        tgt = np.array([prefixes[s][k] for s, k in rows], dtype=np.int64)
        assert tgt.shape[1] == t
        tpos = np.tile(np.arange(1, t + 1, dtype=np.int64), (len(rows), 1))
        sidx = [s for s, _ in rows]
        with O.bf16_storage():
            out = O.decoder_fwd(w, cfg, tgt, tpos, src_np[sidx], enc[sidx])[0][:, -1]
        got = x.float().cpu().numpy()[[s * bm + k for s, k in rows]]
        err = float(np.abs(got - out).max() / np.abs(out).max())
        worst = max(worst, err)
        assert err <= 2e-2, (step, err)
        # the search's choice against the oracle's: log-softmax of the generator's logits (no x_logit_scale,
        # Translator.py:137) on the oracle's decoder output, plus the running beam scores (Beam.py:55-63)
        logits = O.round_bf16(out).astype(np.float64) @ gen.T
        logp = logits - np.log(np.exp(logits - logits.max(1, keepdims=True)).sum(1, keepdims=True)) - logits.max(1, keepdims=True)
        prev_h, tok_h, sc_h = prev.cpu().numpy(), new_tok.cpu().numpy(), new_scores.cpu().numpy()
        at = 0
        for s in range(n):
            if not live[s]:
                continue
            ks = [k for k in range(bm) if prefixes[s][k] is not None]
            cand = np.stack([scores_prev[s, k] + logp[at + i] for i, k in enumerate(ks)])   # [beams][V]
            at += len(ks)
            flat = cand.reshape(-1)
            order = np.argsort(-flat)[:2]
            if flat[order[0]] - flat[order[1]] > 0.05:
                checked += 1
                assert (ks[order[0] // V], order[0] % V) == (int(prev_h[s, 0]), int(tok_h[s, 0])), (step, s)
            new = [prefixes[s][int(prev_h[s, k])] + [int(tok_h[s, k])] for k in range(bm)]
            prefixes[s], scores_prev[s] = new, sc_h[s]
            if tok_h[s, 0] == O.EOS:   # Beam.done: the sentence's state is frozen from here on (Beam.py:74-77)
                live[s] = False
    assert checked >= 30, checked
    print(f"worst relative difference to the bf16-mode oracle over {len(dec.trace)} steps: {worst:.4f}; "
          f"{checked} (step, sentence) choices checked against the oracle's")


def test_full_size_decode_stack_against_op_by_op():
    """The decode configuration at full size — 128 sentences x 5 beams = 640 rows, 150 steps, 6 + 6 layers, V = 35 000,
    source lengths U[12, 42].  (1) Through the one-launch stack replayed from HIP graphs: two runs give the same bits,
    the stack's error word stays 0 (BeamDecoder.run raises otherwise), n-best scores are ordered.  (2) Against the
    op-by-op path of the same model with that path's choices forced onto the stack (so that both see the same 150
    histories): every step's decoder output within 2e-2.  (3) Unforced, the two searches are compared for the record
    only: random weights leave near-ties that a different bf16 summation order breaks differently at some step, after
    which two searches are different searches."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    from specialk_amd.decode import BeamDecoder
    cfg = dict(bench.CONFIGS["cfg2"]); cfg["hi"] = 150
    m, _ = bench.build(cfg, torch.bfloat16, DEV)
    m.eval()
    g = torch.Generator().manual_seed(3)
    s, sp = bench.make_batch(g, 128, 12, 42, cfg["vocab"])
    s, sp = s.to(DEV), sp.to(DEV)
    dec = BeamDecoder(m, 5, 150)
    h1, sc1 = dec.run(s, sp, 3)
    assert dec.used_stack and dec.last_steps == 150
    h2, sc2 = dec.run(s, sp, 3)
    assert h1 == h2 and all(torch.equal(a, b) for a, b in zip(sc1, sc2))
    assert all(len(h) == 3 and all(1 <= len(x) <= 150 for x in h) for h in h1)
    assert all(bool((x[:-1] >= x[1:]).all()) for x in sc1)
    ref = BeamDecoder(m, 5, 150, use_graph=False, use_stack=False)
    ref.trace = []
    h0, _ = ref.run(s, sp, 3)
    assert not ref.used_stack and len(ref.trace) == 150
    got = BeamDecoder(m, 5, 150, use_graph=False, use_stack=True)
    got.trace, got.forced = [], [t[1:] for t in ref.trace]
    hf, _ = got.run(s, sp, 3)
    assert got.used_stack and hf == h0
    worst = 0.0
    for step, (a, b) in enumerate(zip(got.trace, ref.trace)):
        xa, xb = a[0].float(), b[0].float()
        err = float((xa - xb).abs().max() / xb.abs().max())
        worst = max(worst, err)
        assert err <= 2e-2, (step, err)
    same = sum(a[0] == b[0] for a, b in zip(h0, h1))
    print(f"full size: worst relative difference of a step's decoder output over 150 steps {worst:.4f}; unforced, "
          f"{same} / 128 best hypotheses identical over 150 tokens")


def test_error_word_sends_the_batch_to_the_op_by_op_path():
    """A set error word (a member gave up, or a team found itself on two XCDs) must not reach the caller as an
    exception or as wrong tokens: the batch is searched again with one launch per operation."""
    import warnings
    from specialk_amd.decode import BeamDecoder
    model = build(2)
    src, pos = batch(6, 5, 12, 1000, 3)
    ref_dec = BeamDecoder(model, 4, 12, use_stack=False)
    ref_hyp, ref_scores = ref_dec.run(src, pos, n_best=2)
    for code in (1, 2):
        dec = BeamDecoder(model, 4, 12, use_stack=True)
        dec.poke_stack_error = code
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            hyp, scores = dec.run(src, pos, n_best=2)
        assert dec.stack_fallbacks == 1 and not dec.used_stack
        assert any("skhip_decode_stack" in str(x.message) for x in w)
        assert hyp == ref_hyp
        for a, b in zip(scores, ref_scores):
            assert torch.equal(a, b)


def test_teams_report_one_xcd_each():
    """After a launch every team's XCC mask has exactly one bit (its 8 workgroups share an L2) and the error word is 0."""
    from specialk_amd import _lib as L
    from specialk_amd.decode import BeamDecoder
    model = build(1)
    src, pos = batch(40, 5, 9, 1000, 4)   # 200 rows: several teams
    dec = BeamDecoder(model, 5, 4, use_stack=True)
    seen = {}
    real = L.decode_stack

    def spy(*a, **k):
        seen["ws"] = a[12]
        return real(*a, **k)

    L.decode_stack = spy
    try:
        dec.run(src, pos)
    finally:
        L.decode_stack = real
    assert dec.used_stack and dec.stack_fallbacks == 0
    words = seen["ws"][:1024].view(torch.int32).cpu()
    assert int(words[L.STACK_ERR_WORD]) == 0
    masks = words[192:256].view(2, 32)
    used = masks[masks != 0]
    assert used.numel() >= 2
    assert all(bin(int(m)).count("1") == 1 for m in used), masks


def test_slab_generator_search_equals_tiled_generator_search(monkeypatch):
    """The stack handing its rows to the vocabulary-slab generator product in fragment-major order (the default)
    against the same search with the tiled product on row-major rows: the products are equal bit for bit, so the
    hypotheses are the same and the scores differ by the statistics' summation order only."""
    from specialk_amd.decode import BeamDecoder
    m = build(2, vocab=1003)
    src, pos = batch(37, 3, 17, 1003, 4)
    h1, s1 = BeamDecoder(m, 5, 14, use_graph=True, use_stack=True).run(src, pos, 2)
    monkeypatch.setenv("SK_DECODE_GEN_SLAB", "0")
    h0, s0 = BeamDecoder(m, 5, 14, use_graph=True, use_stack=True).run(src, pos, 2)
    assert h1 == h0
    assert max(float((a - b).abs().max()) for a, b in zip(s0, s1)) <= 1e-4


def test_kept_buffers_and_graph_serve_later_batches(monkeypatch):
    """A decoder keeps its buffers and the captured graph per batch shape: three batches of one shape (and one of
    another in between) through ONE decoder give, batch by batch, exactly what a decoder that builds everything
    afresh gives, and what an earlier call returned is not touched by a later one."""
    from specialk_amd.decode import BeamDecoder
    m = build(2)
    batches = [batch(24, 9, 9, 1000, s) for s in (11, 12)] + [batch(24, 5, 5, 1000, 13)] + [batch(24, 9, 9, 1000, 14)]
    kept = BeamDecoder(m, 5, 14, use_graph=True, use_stack=True)
    got = [kept.run(s, p, 2) for s, p in batches]
    snap = [[x.clone() for x in g[1]] for g in got]
    assert len(kept._prep[1]["runs"]) == 2 and all("graph" in b for b in kept._prep[1]["runs"].values())
    monkeypatch.setenv("SK_DECODE_REUSE", "0")
    for (s, p), (h, sc), sn in zip(batches, got, snap):
        h0, s0 = BeamDecoder(m, 5, 14, use_graph=True, use_stack=True).run(s, p, 2)
        assert h == h0
        assert all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(sc, s0, sn))


def test_device_backtrack_equals_host_walk(monkeypatch):
    """skhip_beam_backtrack (a workgroup per sentence walks the back-pointers in LDS) against the host walk with one
    gather per position: same hypotheses for every sentence and rank, with sentences that finish at different steps
    (a copy of the first source tokens would need a trained model: the random one's searches end when the step
    limit does, so lengths are forced through `max_token_seq_len` and n_best = 1, 3, 5)."""
    from specialk_amd.decode import BeamDecoder
    m = build(1)
    src, pos = batch(19, 3, 17, 1000, 8)
    for nb in (1, 3, 5):
        h1, s1 = BeamDecoder(m, 5, 17, use_graph=True, use_stack=True).run(src, pos, nb)
        monkeypatch.setenv("SK_DECODE_BACKTRACK", "0")
        h0, s0 = BeamDecoder(m, 5, 17, use_graph=True, use_stack=True).run(src, pos, nb)
        monkeypatch.delenv("SK_DECODE_BACKTRACK")
        assert h1 == h0 and all(len(x) == nb for x in h1)
    # the kernel alone on a made-up history with unequal lengths
    from specialk_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    n, bm, steps = 7, 4, 23
    ys = torch.randint(5, 90, (steps + 1, n, bm), generator=g, dtype=torch.int32).to(DEV)
    ks = torch.randint(0, bm, (steps, n, bm), generator=g, dtype=torch.int32).to(DEV)
    length = torch.tensor([23, 1, 7, 30, 12, 23, 2], dtype=torch.int32, device=DEV)
    got = L.beam_backtrack(ys, ks, length, n, bm, 3, steps).cpu()
    for s in range(n):
        T = min(int(length[s]), steps)
        for r in range(3):
            b, want = r, [0] * steps
            for j in range(T - 1, -1, -1):
                want[j] = int(ys[j + 1, s, b])
                b = int(ks[j, s, b])
            assert got[s, r].tolist() == want, (s, r)
