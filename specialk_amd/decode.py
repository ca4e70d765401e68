"""Incremental beam-search decoding with a KV cache, beams resident on the GPU.

Computes exactly what Translator.translate_batch of the reference computes
(lib/transformer/Translator.py:55-226 with Beam, lib/transformer/Beam.py:14-110) but:
  * each step runs the decoder on the newest position only; keys/values of earlier positions
    live in per-layer caches that stay in place — a small (row, position) -> cache-row table is
    re-ordered by the beams' back-pointers instead, and the attention kernel follows it
    (the reference re-runs the whole prefix, O(t^2), and re-orders whole tensors);
  * encoder-side keys/values are projected once per sentence and shared by its beams;
  * log-softmax, score accumulation and top-k of ALL sentences are two kernel launches
    (the reference loops over Python Beam objects and syncs with .item());
  * the host reads one flag per step (all sentences finished?) and the token history once at
    the end.
Reference quirks kept: the generator is applied without x_logit_scale (Translator.py:137);
a sentence ends when its TOP beam emits EOS, other beams keep extending past EOS; no length
penalty; step 1 only expands beam 0.
"""
import os
import time
import warnings

import numpy as np
import torch

from . import _lib as L
from . import ops
from .core import constants as Constants


class _StackGaveUp(RuntimeError):
    """The one-launch decoder stack set its error word: the batch's results are not to be used."""


class BeamDecoder:
    def __init__(self, model, beam_size, max_token_seq_len, use_graph=None, use_stack=None):
        self.model = model
        self.bm = int(beam_size)
        self.max_len = int(max_token_seq_len)
        # replay pairs of steps from a captured HIP graph (SK_DECODE_GRAPH=0: launch every step from the host)
        self.use_graph = os.environ.get("SK_DECODE_GRAPH", "1") != "0" if use_graph is None else bool(use_graph)
        # run the decoder stack of a step as ONE launch where the shape allows it (bf16, d_model 512, 8 heads of 64,
        # d_inner 2048, at most 32 x CUs / 8 beam rows); SK_DECODE_STACK=0: always op by op
        self.use_stack = os.environ.get("SK_DECODE_STACK", "1") != "0" if use_stack is None else bool(use_stack)
        self.stack_fallbacks = 0       # batches searched again op by op after the stack kernel's error word was set
        self.poke_stack_error = 0      # tests only: set the error word to this value after the steps
        # debugging aids (host-launched steps only): `trace` receives (decoder output, new_scores, prev, new_tok) of
        # every step; `forced` replaces the search's choice of step i by forced[i] = (new_scores, prev, new_tok)
        self.trace, self.forced = None, None

    _capture_streams = {}

    @classmethod
    def _capture(cls, fn, dev):
        """Capture what fn launches into a graph.  The stream of the capture is kept per device and gets its kernel
        workspace before the capture starts, so that no allocation that outlives the graph comes from the graph's
        private pool."""
        stream = cls._capture_streams.get(dev)
        if stream is None:
            stream = cls._capture_streams[dev] = torch.cuda.Stream(dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            L.workspace()
        torch.cuda.current_stream(dev).wait_stream(stream)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            fn()
        return graph

    @classmethod
    def _capture_pair(cls, step, cur, dev, pairs=1):
        """Capture `pairs` pairs of consecutive steps (history table cur -> cur ^ 1 -> cur) into a graph."""
        def steps():
            for _ in range(pairs):
                step(False, cur)
                step(False, cur ^ 1)
        return cls._capture(steps, dev)

    def _prepared(self, layers, dtype, dev):
        """Weights in the forms a call needs (stacked q | k | v, fragment-major matrices), kept between calls.  They are
        current as long as no optimiser update of this package has run (ops.weights_epoch: its kernel changes parameters
        without touching torch's version counters) and no parameter's version counter has moved (every other way of
        writing a parameter).  SK_DECODE_KEEP_WEIGHTS=0: rebuilt every call."""
        params = [p for l in layers for p in l.parameters()]
        key = (ops.weights_epoch(), dtype, dev, tuple(p._version for p in params))
        held = getattr(self, "_prep", None)
        # (the parameters themselves are held with the key and compared by identity: an id alone could be a new
        # tensor's at a freed one's address)
        if (held is None or held[0] != key or len(held[2]) != len(params) or any(a is not b for a, b in zip(held[2], params))
                or os.environ.get("SK_DECODE_KEEP_WEIGHTS", "1") == "0"):
            held = self._prep = (key, {}, params)
        return held[1]

    def _w(self, p, dtype):
        w = ops.compute_weight(p, dtype)
        return w.reshape(w.shape[0], -1)

    @torch.no_grad()
    def run(self, src_seq, src_pos, n_best=1):
        """Beam search of a batch (Translator.translate_batch, Translator.py:53-171).  The one-launch decoder stack
        reports through an error word when its workgroups could not exchange their slices (a team not resident
        together or not on one XCD): the batch is then searched again with one launch per operation, in this
        process, and self.stack_fallbacks counts it."""
        try:
            return self._run(src_seq, src_pos, n_best, self.use_stack)
        except _StackGaveUp as e:
            self.stack_fallbacks = getattr(self, "stack_fallbacks", 0) + 1
            warnings.warn(f"{e}; searching this batch again with one launch per operation")
            return self._run(src_seq, src_pos, n_best, False)

    def _run(self, src_seq, src_pos, n_best, use_stack):
        model, bm, Tmax = self.model, self.bm, self.max_len
        dec = model.decoder
        dt = dec.compute_dtype
        dev = src_seq.device
        n, Ls = src_seq.shape
        R = n * bm
        layers = list(dec.layer_stack)
        NL = len(layers)
        mha0 = layers[0].slf_attn
        H, dk, dv = mha0.n_head, mha0.d_k, mha0.d_v
        d = dec.tgt_word_emb.weight.shape[1]
        scale = 1.0 / float(mha0.attention.temperature)
        n_pos = dec.position_enc.weight.shape[0]
        if Tmax >= n_pos:
            raise IndexError(f"max_token_seq_len={Tmax} needs a position table of {Tmax + 1} rows, have {n_pos}")

        # SK_DECODE_TIMES=1: wall-clock of the call's parts (with a device synchronisation at each boundary) in self.times
        marks = [] if os.environ.get("SK_DECODE_TIMES") else None

        def mark(name):
            if marks is not None:
                torch.cuda.synchronize(dev)
                marks.append((name, time.perf_counter()))

        mark("start")
        was_training = model.training
        model.eval()
        # Buffers, and the graph captured over them, are kept from call to call with the prepared weights (so: until
        # a parameter changes) for the last two batch shapes: a batch of a known shape then costs no capture (0.6 ms),
        # no zeroing of the caches (1.3 GB at 640 rows) and no allocations.  Kept only for the path that replays
        # graphs of the one-launch stack; SK_DECODE_REUSE=0: everything afresh for every call.
        F = layers[0].pos_ffn.w_1.weight.shape[0]
        # (teams of 8 workgroups x 32 rows, at most 32 teams, all resident together at one workgroup per CU)
        stack = (use_stack and dt == torch.bfloat16 and (d, H, dk, dv, F) == (512, 8, 64, 64, 2048) and NL <= 8
                 and R <= 32 * min(32, torch.cuda.get_device_properties(dev).multi_processor_count // 8))
        prep = self._prepared(layers, dt, dev)
        gen_w = self._w(model.generator.weight, dt)
        V = gen_w.shape[0]
        word_emb, pos_tab = dec.tgt_word_emb.weight.detach(), dec.position_enc.weight.detach()
        reuse = (self.use_graph and stack and self.forced is None and self.trace is None
                 and os.environ.get("SK_DECODE_REUSE", "1") != "0")
        B = {}
        if reuse:
            runs, key = prep.setdefault("runs", {}), (n, Ls, bm, Tmax)
            ptrs = (gen_w.data_ptr(), word_emb.data_ptr(), pos_tab.data_ptr())
            B = runs.get(key)
            if B is None or B["ptrs"] != ptrs:
                while len(runs) >= 2:
                    runs.pop(next(iter(runs)))
                B = runs[key] = {"ptrs": ptrs}

        def buf(name, shape, dtype, fill=None, once=False):
            """The call's tensor `name`: a new one, or the kept one of this batch shape; filled with `fill` (once=True:
            only when new)."""
            t = B.get(name)
            new = t is None
            if new:
                t = B[name] = torch.empty(shape, dtype=dtype, device=dev)
            if fill is not None and (new or not once):
                t.fill_(fill)
            return t

        src_key_pad = buf("src_key_pad", (n, Ls), torch.uint8)
        src_key_pad.copy_(src_seq.eq(Constants.PAD))

        # the encoder pass and the encoder-side K/V: once per sentence, shared by its beams.  (Capturing these ≈ 50
        # launches per batch shape as well was tried: no faster — 0.4427 against 0.4419 ms per step of a 150-step batch,
        # 14.7 k against 14.9 k sentences/s on 28-token outputs.)
        enc_out, *_ = model.encoder(src_seq, src_pos)
        enc2d = enc_out.reshape(n * Ls, d)
        if dt == torch.bfloat16 and dk == dv and NL > 1 and os.environ.get("SK_DECODE_KV_FUSED", "1") != "0":
            # all layers' k | v projections as ONE product against the stacked weights, then one copy that takes the
            # [row][layer, k | v][512] result apart into per-layer matrices (12 split-K products of 512 columns on a
            # few thousand rows: 0.44 ms per batch; this way: 0.1)
            cross = buf("crossKV", (2 * NL, n, Ls, H * dk), dt)
            crossK, crossV = cross[0::2], cross[1::2]
            mods = [l.enc_attn for l in layers]
            Wkv = ops.derived_weight("cat", tuple(w for a in mods for w in (a.w_ks.weight, a.w_vs.weight)))
            bkv = ops.derived_weight("bias", tuple(w for a in mods for w in (a.w_ks.bias, a.w_vs.bias)))
            wide = L.gemm(L.GEMM_NT, enc2d, Wkv, n * Ls, 2 * NL * H * dk, d, d, d, bias=bkv)
            cross.view(2 * NL, n * Ls, H * dk).copy_(wide.view(n * Ls, 2 * NL, H * dk).transpose(0, 1))
        else:
            crossK = buf("crossK", (NL, n, Ls, H * dk), dt)
            crossV = buf("crossV", (NL, n, Ls, H * dv), dt)
            for li, layer in enumerate(layers):
                a = layer.enc_attn
                L.gemm(L.GEMM_NT, enc2d, self._w(a.w_ks.weight, dt), n * Ls, H * dk, d, d, d, out=crossK[li], ldc=H * dk,
                       bias=a.w_ks.bias.detach())
                L.gemm(L.GEMM_NT, enc2d, self._w(a.w_vs.weight, dt), n * Ls, H * dv, d, d, d, out=crossV[li], ldc=H * dv,
                       bias=a.w_vs.bias.detach())

        # self-attention caches: position t of the hypothesis that occupies row r at step t is written to
        # cache row r and never moved; hist[r][j] names the cache row of position j of row r's history
        # and is what the beam re-ordering permutes (double buffered).  Time axis rounded up to 16.  (Kept caches
        # are not cleared: positions past a row's length are masked where they are read, and what lies there are an
        # earlier batch's finite keys and values.)
        Tc = (Tmax + 15) // 16 * 16
        Kc = buf("Kc", (NL, R, Tc, H * dk), dt, 0, once=True)
        Vc = buf("Vc", (NL, R, Tc, H * dv), dt, 0, once=True)
        i32 = dict(dtype=torch.int32, device=dev)
        pad_hist = buf("pad_hist", (R, Tc), torch.uint8, 0)
        hist = [buf("hist0", (R, Tc), torch.int32, 0), buf("hist1", (R, Tc), torch.int32, 0)]
        if "row_ids" not in B:
            B["row_ids"] = torch.arange(R, **i32)
        hist[0][:, 0] = B["row_ids"]

        # Beam state and the inputs of the next step live on the device and are advanced by one kernel per step
        # (skhip_beam_update), including the step counter (cur_len): a step's launches do not depend on the
        # step number, so pairs of steps (the history table is double buffered) are captured in a graph once
        # per batch shape and replayed.
        scores = buf("scores", (n, bm), torch.float32, 0)
        new_scores = buf("new_scores", (n, bm), torch.float32)
        prev, new_tok = buf("prev", (n, bm), torch.int32), buf("new_tok", (n, bm), torch.int32)
        ys = buf("ys", (Tmax + 1, n, bm), torch.int32, 0)
        ks = buf("ks", (Tmax, n, bm), torch.int32, 0)
        done, n_done = buf("done", (n,), torch.int32, 0), buf("n_done", (1,), torch.int32, 0)
        length = buf("length", (n,), torch.int32, Tmax)
        cur_len = buf("cur_len", (R,), torch.int32, 1)
        tok = buf("tok", (R,), torch.int64, Constants.SOS)
        pos = buf("pos", (R,), torch.int64, 1)
        row_pad = buf("row_pad", (R,), torch.uint8, 0)

        # q | k | v of the self-attention in one projection: weights stacked, all layers in one copy, and kept from
        # call to call while the weights stay what they were (see _prepared)
        nq, nk, nv = H * dk, H * dk, H * dv
        if "qkv_w" not in prep:
            prep["qkv_w"] = torch.cat([self._w(p, dt) for l in layers for p in (l.slf_attn.w_qs.weight, l.slf_attn.w_ks.weight,
                                                                               l.slf_attn.w_vs.weight)], 0).view(NL, nq + nk + nv, d)
            prep["qkv_b"] = torch.cat([p.detach() for l in layers for p in (l.slf_attn.w_qs.bias, l.slf_attn.w_ks.bias,
                                                                            l.slf_attn.w_vs.bias)]).float().view(NL, nq + nk + nv)
        qkv_w, qkv_b = prep["qkv_w"], prep["qkv_b"]

        def proj(x, lin, out=None, ldc=None, relu=False):
            w = self._w(lin.weight, dt)
            # a (R, 512) output is a handful of tiles: split-K only pays for the long contraction of w_2,
            # elsewhere its slab write + reduce launch costs more than it saves
            return L.gemm(L.GEMM_NT, x, w, R, w.shape[0], w.shape[1], x.stride(0), w.shape[1], out=out, ldc=ldc,
                          bias=lin.bias.detach(), epilogue=L.EPI_RELU if relu else 0, split_k=w.shape[1] >= 2048)

        def add_ln(o, x, ln):
            y, _, _ = L.add_ln_fwd(o, x, ln.weight.detach(), ln.bias.detach(), row_pad, ln.eps, 0.0, 0)
            return y

        if stack:
            f32 = lambda p: p.detach().float().contiguous()

            def frag_major(mats):
                """The kernel reads its weight matrices fragment-major, [rows / 16][K / 32][64 lanes][8]: matrices of
                one shape are permuted together (one copy kernel per shape)."""
                W = mats if torch.is_tensor(mats) else torch.stack(mats)
                n_, N_, K_ = W.shape
                return W.view(n_, N_ // 16, 16, K_ // 32, 4, 8).permute(0, 1, 3, 4, 2, 5).contiguous()

            if "fm_qkv" not in prep:
                prep["fm_qkv"] = frag_major(qkv_w)
                prep["fm_sq"] = frag_major([self._w(p, dt) for l in layers for p in (l.slf_attn.fc.weight, l.enc_attn.w_qs.weight,
                                                                                      l.enc_attn.fc.weight)])
                prep["fm_w1"] = frag_major([self._w(l.pos_ffn.w_1.weight, dt) for l in layers])
                prep["fm_w2"] = frag_major([self._w(l.pos_ffn.w_2.weight, dt) for l in layers])
            fm_qkv, fm_sq, fm_w1, fm_w2 = prep["fm_qkv"], prep["fm_sq"], prep["fm_w1"], prep["fm_w2"]
            rows = B.get("stack_rows", [])
            for li, l in enumerate(layers if not rows else []):
                s_, c_, f_ = l.slf_attn, l.enc_attn, l.pos_ffn
                row = [fm_qkv[li], qkv_b[li], fm_sq[3 * li], f32(s_.fc.bias), f32(s_.layer_norm.weight),
                       f32(s_.layer_norm.bias), fm_sq[3 * li + 1], f32(c_.w_qs.bias), fm_sq[3 * li + 2],
                       f32(c_.fc.bias), f32(c_.layer_norm.weight), f32(c_.layer_norm.bias), fm_w1[li],
                       f32(f_.w_1.bias), fm_w2[li], f32(f_.w_2.bias), f32(f_.layer_norm.weight),
                       f32(f_.layer_norm.bias), Kc[li], Vc[li], crossK[li], crossV[li]]
                rows.append(row)
            if "stack_table" not in B:   # (the rows stay referenced with the table of their addresses)
                B["stack_rows"], B["stack_table"] = rows, L.decode_stack_table(rows)[0]
            stack_table = B["stack_table"]
            # (zeroed for every call: a call's first launch has parity 0 whatever the last launch of the call before had)
            stack_ws = buf("stack_ws", (L.decode_stack_ws_bytes(R),), torch.uint8, 0)
            ln_eps = float(layers[0].slf_attn.layer_norm.eps)
        self.used_stack = bool(stack)
        use_stats = (dt == torch.bfloat16 and d % 8 == 0 and V <= 65536 and os.environ.get("SK_DECODE_TOPK_STATS", "1") != "0")
        # the generator product by vocabulary slabs (one round of the chip, the weight read once) when the stack
        # kernel can hand it the rows in MFMA operand order; its statistics chunks are 48 columns wide
        gen_slab = bool(stack) and use_stats and V <= 1024 * L.generator_chunk_cols() and os.environ.get("SK_DECODE_GEN_SLAB", "1") != "0"
        chunk = L.generator_chunk_cols() if gen_slab else 64
        if stack:
            stack_out = buf("stack_out", ((R + 15) // 16 * 16, d), dt, 0, once=True)
        if use_stats:
            logits_buf = buf("logits", (R, V), torch.float32)
            gen_stats = buf("gen_stats", (R, (V + chunk - 1) // chunk, 2), torch.float32)
        # (tests that force or trace a step's choices sit between the top-k and the update: they keep the separate launches)
        fused_step = (use_stats and self.forced is None and self.trace is None
                      and os.environ.get("SK_DECODE_FUSED_STEP", "1") != "0")
        step_no = [0]

        def step(first, cur):
            """One decoder step on the newest position of every beam, then the beam update; reads the history
            table hist[cur] and leaves the re-parented one in hist[cur ^ 1]."""
            if stack:
                L.decode_stack(stack_table, NL, tok, pos, word_emb, pos_tab, cur_len, row_pad, pad_hist, hist[cur],
                               src_key_pad, stack_out, stack_ws, R, bm, Tc, Ls, d, H, dk, F, scale, ln_eps, cur,
                               L.ROWS_FRAGMENT_MAJOR if gen_slab else L.ROWS_PLAIN)
                return finish_step(stack_out if gen_slab else stack_out[:R], first, cur)
            x = L.embed_fwd(tok, pos, word_emb, pos_tab, dt)
            for li, layer in enumerate(layers):
                a = layer.slf_attn
                qkv = L.gemm(L.GEMM_NT, x, qkv_w[li], R, nq + nk + nv, d, x.stride(0), d, bias=qkv_b[li], split_k=False)
                att = torch.empty((R, H * dv), dtype=dt, device=dev)
                # the kernel takes the newest position's key / value from the projection and appends them to the caches
                L.decode_attn(qkv, Kc[li], Vc[li], att, R, H, 0, Tc, dk, dv, nq + nk + nv, H * dv, scale, kv_len=cur_len,
                              key_pad=pad_hist, hist_row=hist[cur], new_k=qkv[:, nq:], new_v=qkv[:, nq + nk:],
                              ld_new=nq + nk + nv)
                x = add_ln(proj(att, a.fc), x, a.layer_norm)
                a = layer.enc_attn
                q = proj(x, a.w_qs)
                L.decode_attn(q, crossK[li], crossV[li], att, R, H, Ls, Ls, dk, dv, H * dk, H * dv, scale,
                              key_pad=src_key_pad, kv_row_div=bm)
                x = add_ln(proj(att, a.fc), x, a.layer_norm)
                f = layer.pos_ffn
                hbuf = proj(x, f.w_1, relu=True)
                x = add_ln(proj(hbuf, f.w_2), x, f.layer_norm)
            return finish_step(x, first, cur)

        def finish_step(x, first, cur):
            if use_stats:
                # bf16 models: the generator product also leaves each 64-column chunk's maximum and sum of exponentials;
                # the top-k then reads those and the few chunks that can hold a best token, not every row twice
                if gen_slab:
                    L.generator_rowstats(x, gen_w, logits_buf, gen_stats, L.ROWS_FRAGMENT_MAJOR)
                else:
                    L.gemm(L.GEMM_NT, x, gen_w, R, V, d, d, d, out=logits_buf, ldc=V, aux=gen_stats, epilogue=L.EPI_ROWSTATS,
                           split_k=False)
                if fused_step:  # top-k, merge and state update in one launch
                    L.beam_step(logits_buf, gen_stats, (new_scores, prev, new_tok), scores, ys, ks, done, length, n_done,
                                cur_len, tok, pos, row_pad, pad_hist, hist[cur], hist[cur ^ 1], n, bm, first, Tc, Tmax,
                                Constants.PAD, Constants.EOS, chunk_cols=chunk)
                    step_no[0] += 1
                    return
                L.beam_topk_stats(logits_buf, gen_stats, scores, n, bm, first, out=(new_scores, prev, new_tok), chunk_cols=chunk)
            else:  # fp32 models keep expf and the reference's bits
                logits = L.gemm(L.GEMM_NT, x, gen_w, R, V, d, d, d, out_dtype=torch.float32)
                L.beam_topk(logits, scores, n, bm, first_step=first, normalize=2 if dt == torch.bfloat16 else 1,
                            out=(new_scores, prev, new_tok))
            if self.forced is not None:
                for dst, src in zip((new_scores, prev, new_tok), self.forced[step_no[0]]):
                    dst.copy_(src)
            if self.trace is not None:
                self.trace.append(((L.plain_rows(x, R) if gen_slab and stack else x).clone(), new_scores.clone(), prev.clone(),
                                   new_tok.clone()))
            step_no[0] += 1
            # finished sentences keep their final beam state (their Beam is no longer advanced)
            L.beam_update(new_scores, prev, new_tok, scores, ys, ks, done, length, n_done, cur_len, tok, pos, row_pad,
                          pad_hist, hist[cur], hist[cur ^ 1], n, bm, Tc, Tmax, Constants.PAD, Constants.EOS)

        mark("encoder + set-up")
        step(True, 0)  # step 1 expands beam 0 only (Beam.py:51-54)
        steps, cur = 1, 1
        graph = None
        # Graphs of four steps (two for very short runs): the host looks at the finished-sentence count every fourth step
        # only — a step past the end changes nothing (finished sentences keep their state) and costs less than a
        # pipeline drain per step (a blocking read leaves the step stream idle for some 40 us; with steps of 0.5 ms
        # and runs of 40 .. 150 steps, four is about where drains and steps past the end cost the same).
        per = 4 if Tmax >= 6 else 2
        if self.use_graph and Tmax >= 3:
            graph = B.get("graph")   # (captured over this batch shape's kept buffers by an earlier call)
            if graph is None:
                graph = B["graph"] = self._capture_pair(step, cur, dev, per // 2)
        mark("first step + graph capture")
        next_look = steps + 4
        while steps < Tmax:
            if graph is not None and steps + per <= Tmax:
                graph.replay()
                steps += per
            else:
                step(False, cur)
                cur ^= 1
                steps += 1
            if steps >= next_look or steps == Tmax:
                finished = int(n_done.item())
                if finished == n:
                    break
                next_look = steps + 4
        mark("steps")
        del graph
        if not reuse:
            B.clear()
        self.last_steps = steps
        if was_training:
            model.train()
        if stack:
            if self.poke_stack_error:   # tests: as if a member had given up
                stack_ws[4 * L.STACK_ERR_WORD:4 * L.STACK_ERR_WORD + 4].view(torch.int32).fill_(int(self.poke_stack_error))
            err = int(stack_ws[4 * L.STACK_ERR_WORD:4 * L.STACK_ERR_WORD + 4].view(torch.int32).item())
            if err != 0:
                raise _StackGaveUp("skhip_decode_stack: " + ("a team's workgroups ran on more than one XCD" if err == 2 else
                                                             "a team member stopped waiting for its team"))

        # hypotheses: the back-pointer walk of Beam.get_hypothesis (Beam.py:103-110) for all sentences and ranks, on the
        # device when a sentence's history fits its workgroup's LDS (one launch, one small copy), else on the host
        # with one gather per position
        nb = min(n_best, bm)
        scores_h = scores
        if 8 * steps * bm <= 64 * 1024 and os.environ.get("SK_DECODE_BACKTRACK", "1") != "0":
            toks = L.beam_backtrack(ys, ks, length, n, bm, nb, steps).cpu().numpy()   # [n][nb][steps]
            T_all = np.minimum(length.cpu().numpy(), steps).astype(np.int64)
        else:
            ys_h, ks_h = ys[: steps + 1].cpu().numpy(), ks[:steps].cpu().numpy()
            T_all = np.minimum(length.cpu().numpy(), steps).astype(np.int64)  # [n]
            T_max = int(T_all.max()) if n else 0
            ys_f, ks_f = ys_h.reshape(steps + 1, n * bm), ks_h.reshape(steps, n * bm)
            base = (np.arange(n, dtype=np.int64) * bm)[:, None]
            k = np.tile(np.arange(nb, dtype=np.int64)[None, :], (n, 1))   # [n][nb]: beam of rank k0 at the last step
            toks = np.zeros((max(T_max, 1), n, nb), dtype=np.int64)
            all_live = bool((T_all == T_max).all())
            for j in range(T_max - 1, -1, -1):
                at = base + k
                toks[j] = ys_f[j + 1].take(at)
                k = ks_f[j].take(at) if all_live else np.where((j < T_all)[:, None], ks_f[j].take(at), k)
            toks = toks.transpose(1, 2, 0)                                # [n][nb][T]: sentences shorter than T_max keep their prefix
        all_hyp = [[toks[s, r, :T_all[s]].tolist() for r in range(nb)] for s in range(n)]
        all_scores = list((scores_h.clone() if reuse else scores_h)[:, :n_best].unbind(0))   # (kept buffers are the next call's)
        mark("hypotheses")
        if marks is not None:
            self.times = {b[0]: b[1] - a[1] for a, b in zip(marks, marks[1:])}
        return all_hyp, all_scores
