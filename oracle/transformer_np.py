"""CPU oracle: a numpy restatement of the reference's Transformer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under specialk_amd/ imports this module; it is used by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker and the
CPU baseline, never as a product code path.

Parity status: PINNED.  Every function below is checked in tests/test_oracle.py against
golden vectors produced by running the reference itself (tools/make_golden.py, fixtures in
tests/golden/).

All functions are plain numpy, forward and hand-derived backward, computing in `dtype`
(float32 to mimic the reference, float64 to act as a high-precision arbiter).  Weights are
dicts keyed by the reference's state_dict names.  Citations are to files under
/root/reference/base/.
"""
import numpy as np

import contextlib

PAD, UNK, SOS, EOS, BLO = 0, 1, 2, 3, 4  # core/constants.py:6-10
LN_EPS = 1e-5                             # nn.LayerNorm default used by lib/transformer/SubLayers.py:30,75

# ----------------------------------------------------------------------------- bf16 storage mode
# The reference computes in fp32 throughout.  The accelerated path can keep activations, activation gradients
# and the matrices' working copies in bfloat16 (fp32 accumulation, fp32 statistics, fp32 parameter gradients).
# Inside `with bf16_storage():` the functions below round to bfloat16 (nearest even) at exactly those points —
# every tensor such a path writes to memory in bf16 — and are otherwise unchanged, so the result is what the
# reference's arithmetic gives under that storage format: the yardstick for a bf16 run.  Outside the context
# nothing is rounded and the module is the fp32 / fp64 restatement that the golden vectors pin.
_bf16 = False


@contextlib.contextmanager
def bf16_storage():
    global _bf16
    old, _bf16 = _bf16, True
    try:
        yield
    finally:
        _bf16 = old


def round_bf16(x):
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    u = x32.view(np.uint32)
    r = (u + (((u >> 16) & 1) + 0x7FFF)) & np.uint32(0xFFFF0000)
    out = np.where(np.isnan(x32), x32, r.view(np.float32))
    return out.astype(x.dtype) if isinstance(x, np.ndarray) else out


def _rt(x):
    """Round a stored activation / activation gradient to bf16 in bf16-storage mode."""
    return round_bf16(x) if _bf16 else x


def _wt(w, key):
    """A matrix as the GEMMs read it: its bf16 working copy in bf16-storage mode."""
    return round_bf16(w[key]) if _bf16 else w[key]



# ----------------------------------------------------------------------------- data
def collate(insts):
    """core/dataset.py:21-46 — pad to batch max with PAD; pos = index+1 on non-pad, else 0."""
    max_len = max(len(x) for x in insts)
    seq = np.array([list(x) + [PAD] * (max_len - len(x)) for x in insts], dtype=np.int64)
    pos = np.array([[i + 1 if w != PAD else 0 for i, w in enumerate(row)] for row in seq], dtype=np.int64)
    return seq, pos


def sinusoid_table(n_position, d_hid, padding_idx=None):
    """lib/transformer/Models.py:26-46 — float64 angles, sin on even / cos on odd columns."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    if padding_idx is not None:
        table[padding_idx] = 0.0
    return table.astype(np.float32)


# ----------------------------------------------------------------------------- primitives
def linear_fwd(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def linear_bwd(x, w, dy, has_bias=True):
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    dx = (dy2 @ w).reshape(x.shape)
    dw = dy2.T @ x2
    db = dy2.sum(0) if has_bias else None
    return dx, dw, db


def layer_norm_fwd(z, gamma, beta, eps=LN_EPS):
    mean = z.mean(-1, keepdims=True)
    var = ((z - mean) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (z - mean) * rstd
    return xhat * gamma + beta, (xhat, rstd)


def layer_norm_bwd(dy, gamma, cache):
    xhat, rstd = cache
    dxhat = dy * gamma
    dz = rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    dgamma = (dy * xhat).reshape(-1, dy.shape[-1]).sum(0)
    dbeta = dy.reshape(-1, dy.shape[-1]).sum(0)
    return dz, dgamma, dbeta


def sdpa_fwd(q, k, v, mask, temperature):
    """lib/transformer/Modules.py:17-28 in eval mode (dropout off).
    q (n,Lq,dk) k (n,Lk,dk) v (n,Lk,dv); mask bool (n,Lq,Lk), True = masked with -inf.
    A fully masked row yields NaN exactly as torch's softmax does."""
    s = np.matmul(q, k.transpose(0, 2, 1)) / temperature
    if mask is not None:
        s = np.where(mask, -np.inf, s)
    with np.errstate(invalid="ignore"):
        m = s.max(-1, keepdims=True)
        e = np.exp(s - m)
        p = e / e.sum(-1, keepdims=True)
        if _bf16:  # the un-normalised probabilities are the bf16 operand of P.V; the row sum stays fp32
            return np.matmul(round_bf16(e), v) / e.sum(-1, keepdims=True), p
    return np.matmul(p, v), p


def sdpa_bwd(q, k, v, p, dout, temperature, o=None):
    dv = np.matmul(_rt(p).transpose(0, 2, 1), dout)
    dp = np.matmul(dout, v.transpose(0, 2, 1))
    # bf16 storage: the row term comes from the stored (rounded) output, sum_j dO_j O_j, as a fused kernel has it
    delta = (dout * o).sum(-1, keepdims=True) if (_bf16 and o is not None) else (dp * p).sum(-1, keepdims=True)
    ds = _rt(p * (dp - delta) / temperature)
    dq = np.matmul(ds, k)
    dk = np.matmul(ds.transpose(0, 2, 1), q)
    return dq, dk, dv


# ----------------------------------------------------------------------------- sublayers
def _split_heads(x, n_head, dh):
    B, L, _ = x.shape  # SubLayers.py:46-52: (B,L,h*dh) -> (h*B, L, dh), head-major
    return x.reshape(B, L, n_head, dh).transpose(2, 0, 1, 3).reshape(n_head * B, L, dh)


def _merge_heads(x, n_head, B):
    _, L, dh = x.shape  # SubLayers.py:57-60
    return x.reshape(n_head, B, L, dh).transpose(1, 2, 0, 3).reshape(B, L, n_head * dh)


def mha_fwd(w, pre, q_in, kv_in, mask, n_head, d_k, d_v):
    """lib/transformer/SubLayers.py:37-65 (eval mode).  mask bool (B,Lq,Lk)."""
    B = q_in.shape[0]
    q = _split_heads(_rt(linear_fwd(q_in, _wt(w, pre + "w_qs.weight"), w[pre + "w_qs.bias"])), n_head, d_k)
    k = _split_heads(_rt(linear_fwd(kv_in, _wt(w, pre + "w_ks.weight"), w[pre + "w_ks.bias"])), n_head, d_k)
    v = _split_heads(_rt(linear_fwd(kv_in, _wt(w, pre + "w_vs.weight"), w[pre + "w_vs.bias"])), n_head, d_v)
    mask_h = np.tile(mask, (n_head, 1, 1))  # SubLayers.py:54
    o, p = sdpa_fwd(q, k, v, mask_h, np.power(d_k, 0.5))
    o = _rt(o)
    merged = _merge_heads(o, n_head, B)
    fc = _rt(linear_fwd(merged, _wt(w, pre + "fc.weight"), w[pre + "fc.bias"]))
    out, ln_cache = layer_norm_fwd(fc + q_in, w[pre + "layer_norm.weight"], w[pre + "layer_norm.bias"])
    cache = (q_in, kv_in, q, k, v, p, merged, ln_cache, o)
    return _rt(out), p, cache


def mha_bwd(w, pre, dout, cache, n_head, d_k, d_v, grads):
    q_in, kv_in, q, k, v, p, merged, ln_cache, o = cache
    B = q_in.shape[0]
    dz, dg, db = layer_norm_bwd(dout, w[pre + "layer_norm.weight"], ln_cache)
    _acc(grads, pre + "layer_norm.weight", dg)
    _acc(grads, pre + "layer_norm.bias", db)
    _acc(grads, pre + "fc.bias", dz.reshape(-1, dz.shape[-1]).sum(0))  # summed before dz is stored
    dz = _rt(dz)
    dmerged, dw, _ = linear_bwd(merged, _wt(w, pre + "fc.weight"), dz, has_bias=False)
    _acc(grads, pre + "fc.weight", dw)
    do = _split_heads(_rt(dmerged), n_head, d_v)
    dq, dk, dv = sdpa_bwd(q, k, v, p, do, np.power(d_k, 0.5), o)
    dq, dk, dv = (_rt(_merge_heads(g, n_head, B)) for g in (dq, dk, dv))
    dq_in, dw, dbias = linear_bwd(q_in, _wt(w, pre + "w_qs.weight"), dq)
    _acc(grads, pre + "w_qs.weight", dw)
    _acc(grads, pre + "w_qs.bias", dbias)
    dk_in, dw, dbias = linear_bwd(kv_in, _wt(w, pre + "w_ks.weight"), dk)
    _acc(grads, pre + "w_ks.weight", dw)
    _acc(grads, pre + "w_ks.bias", dbias)
    dv_in, dw, dbias = linear_bwd(kv_in, _wt(w, pre + "w_vs.weight"), dv)
    _acc(grads, pre + "w_vs.weight", dw)
    _acc(grads, pre + "w_vs.bias", dbias)
    # residual = q_in (SubLayers.py:44,63).  bf16 storage: the input gradient of the stacked projections and the
    # residual branch are summed in fp32 and stored once (self-attention: all three projections share the input)
    if _bf16 and kv_in is q_in:
        return _rt(dq_in + dk_in + dv_in + dz), 0
    return _rt(dq_in + dz), _rt(dk_in + dv_in)


def ffn_fwd(w, pre, x):
    """lib/transformer/SubLayers.py:78-85 — Conv1d(k=1) weights are (out,in,1)."""
    w1, w2 = _wt(w, pre + "w_1.weight")[:, :, 0], _wt(w, pre + "w_2.weight")[:, :, 0]
    h = _rt(np.maximum(linear_fwd(x, w1, w[pre + "w_1.bias"]), 0))
    y = _rt(linear_fwd(h, w2, w[pre + "w_2.bias"]))
    out, ln_cache = layer_norm_fwd(y + x, w[pre + "layer_norm.weight"], w[pre + "layer_norm.bias"])
    return _rt(out), (x, h, ln_cache)


def ffn_bwd(w, pre, dout, cache, grads):
    x, h, ln_cache = cache
    w1, w2 = _wt(w, pre + "w_1.weight")[:, :, 0], _wt(w, pre + "w_2.weight")[:, :, 0]
    dz, dg, db = layer_norm_bwd(dout, w[pre + "layer_norm.weight"], ln_cache)
    _acc(grads, pre + "layer_norm.weight", dg)
    _acc(grads, pre + "layer_norm.bias", db)
    _acc(grads, pre + "w_2.bias", dz.reshape(-1, dz.shape[-1]).sum(0))  # summed before dz is stored
    dz = _rt(dz)
    dh, dw2, _ = linear_bwd(h, w2, dz, has_bias=False)
    _acc(grads, pre + "w_2.weight", dw2[:, :, None])
    dh = _rt(dh * (h > 0))
    dx, dw1, db1 = linear_bwd(x, w1, dh)
    _acc(grads, pre + "w_1.weight", dw1[:, :, None])
    _acc(grads, pre + "w_1.bias", db1)
    return _rt(dx + dz)


def _acc(grads, key, val):
    if grads is None or val is None:
        return
    grads[key] = grads[key] + val if key in grads else val


# ----------------------------------------------------------------------------- layers
def enc_layer_fwd(w, pre, x, keypad, cfg):
    """lib/transformer/Layers.py:18-27.  keypad bool (B,L): True where the token is PAD."""
    n_head, d_k, d_v = cfg
    non_pad = (~keypad)[:, :, None].astype(x.dtype)
    mask = np.broadcast_to(keypad[:, None, :], (x.shape[0], x.shape[1], x.shape[1]))
    a, p, c1 = mha_fwd(w, pre + "slf_attn.", x, x, mask, n_head, d_k, d_v)
    a = a * non_pad
    f, c2 = ffn_fwd(w, pre + "pos_ffn.", a)
    f = f * non_pad
    return f, p, (c1, c2, non_pad)


def enc_layer_bwd(w, pre, dout, cache, cfg, grads):
    n_head, d_k, d_v = cfg
    c1, c2, non_pad = cache
    da = ffn_bwd(w, pre + "pos_ffn.", dout * non_pad, c2, grads)
    dq, dkv = mha_bwd(w, pre + "slf_attn.", da * non_pad, c1, n_head, d_k, d_v, grads)
    return dq + dkv


def causal_pad_mask(keypad):
    """lib/transformer/Models.py:60-69,173-175 — key-pad OR strictly-upper-triangular."""
    B, L = keypad.shape
    sub = np.triu(np.ones((L, L), dtype=bool), k=1)
    return keypad[:, None, :] | sub[None]


def dec_layer_fwd(w, pre, x, mem, keypad, src_keypad, cfg):
    """lib/transformer/Layers.py:39-60."""
    n_head, d_k, d_v = cfg
    B, Lt, _ = x.shape
    non_pad = (~keypad)[:, :, None].astype(x.dtype)
    a, p1, c1 = mha_fwd(w, pre + "slf_attn.", x, x, causal_pad_mask(keypad), n_head, d_k, d_v)
    a = a * non_pad
    cross = np.broadcast_to(src_keypad[:, None, :], (B, Lt, mem.shape[1]))
    b, p2, c2 = mha_fwd(w, pre + "enc_attn.", a, mem, cross, n_head, d_k, d_v)
    b = b * non_pad
    f, c3 = ffn_fwd(w, pre + "pos_ffn.", b)
    f = f * non_pad
    return f, p1, p2, (c1, c2, c3, non_pad)


def dec_layer_bwd(w, pre, dout, cache, cfg, grads):
    n_head, d_k, d_v = cfg
    c1, c2, c3, non_pad = cache
    db = ffn_bwd(w, pre + "pos_ffn.", dout * non_pad, c3, grads)
    da, dmem = mha_bwd(w, pre + "enc_attn.", db * non_pad, c2, n_head, d_k, d_v, grads)
    dq, dkv = mha_bwd(w, pre + "slf_attn.", da * non_pad, c1, n_head, d_k, d_v, grads)
    return dq + dkv, dmem


# ----------------------------------------------------------------------------- model
class Config:
    def __init__(self, n_layers, n_head, d_k, d_v, d_model, d_inner, tied_proj=False, x_logit_scale=None):
        self.n_layers, self.n_head, self.d_k, self.d_v = n_layers, n_head, d_k, d_v
        self.d_model, self.d_inner, self.tied_proj = d_model, d_inner, tied_proj
        # lib/transformer/Models.py:259-264
        self.x_logit_scale = (d_model ** -0.5 if tied_proj else 1.0) if x_logit_scale is None else x_logit_scale

    @property
    def attn(self):
        return (self.n_head, self.d_k, self.d_v)


def cast_weights(w, dtype):
    return {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in w.items()}


def encoder_fwd(w, cfg, src_seq, src_pos):
    """lib/transformer/Models.py:108-129."""
    keypad = src_seq == PAD
    x = _rt(w["encoder.src_word_emb.weight"][src_seq] + w["encoder.position_enc.weight"][src_pos])
    caches, attns = [], []
    for i in range(cfg.n_layers):
        x, p, c = enc_layer_fwd(w, f"encoder.layer_stack.{i}.", x, keypad, cfg.attn)
        caches.append(c)
        attns.append(p)
    return x, attns, caches


def encoder_bwd(w, cfg, src_seq, dout, caches, grads):
    for i in reversed(range(cfg.n_layers)):
        dout = enc_layer_bwd(w, f"encoder.layer_stack.{i}.", dout, caches[i], cfg.attn, grads)
    _embed_bwd(grads, "encoder.src_word_emb.weight", w["encoder.src_word_emb.weight"].shape, src_seq, dout)


def _embed_bwd(grads, key, shape, seq, dx):
    if grads is None:
        return
    g = np.zeros(shape, dtype=dx.dtype)
    np.add.at(g, seq.reshape(-1), dx.reshape(-1, dx.shape[-1]))
    g[PAD] = 0  # nn.Embedding(padding_idx=PAD), Models.py:92-94,151-153
    _acc(grads, key, g)


def decoder_fwd(w, cfg, tgt_seq, tgt_pos, src_seq, enc_out):
    """lib/transformer/Models.py:167-199."""
    keypad, src_keypad = tgt_seq == PAD, src_seq == PAD
    x = _rt(w["decoder.tgt_word_emb.weight"][tgt_seq] + w["decoder.position_enc.weight"][tgt_pos])
    caches, a1s, a2s = [], [], []
    for i in range(cfg.n_layers):
        x, p1, p2, c = dec_layer_fwd(w, f"decoder.layer_stack.{i}.", x, enc_out, keypad, src_keypad, cfg.attn)
        caches.append(c)
        a1s.append(p1)
        a2s.append(p2)
    return x, a1s, a2s, caches


def decoder_bwd(w, cfg, tgt_seq, dout, caches, grads):
    dmem = 0
    for i in reversed(range(cfg.n_layers)):
        dout, dm = dec_layer_bwd(w, f"decoder.layer_stack.{i}.", dout, caches[i], cfg.attn, grads)
        dmem = _rt(dmem + dm)  # autograd adds the layers' contributions as stored tensors
    _embed_bwd(grads, "decoder.tgt_word_emb.weight", w["decoder.tgt_word_emb.weight"].shape, tgt_seq, dout)
    return dmem


def transformer_fwd(w, cfg, src_seq, src_pos, tgt_seq, tgt_pos):
    """lib/transformer/Models.py:273-280 — drops the last target column, scales logits."""
    tgt_seq, tgt_pos = tgt_seq[:, :-1], tgt_pos[:, :-1]
    enc_out, _, enc_caches = encoder_fwd(w, cfg, src_seq, src_pos)
    dec_out, _, _, dec_caches = decoder_fwd(w, cfg, tgt_seq, tgt_pos, src_seq, enc_out)
    logits = _rt((dec_out @ _wt(w, "generator.weight").T) * cfg.x_logit_scale)
    return logits, (src_seq, tgt_seq, enc_caches, dec_caches, dec_out)


def transformer_bwd(w, cfg, dlogits, cache):
    """Returns grads keyed like the state_dict.  With tied weights the caller sums the
    aliases (see tie_grads)."""
    src_seq, tgt_seq, enc_caches, dec_caches, dec_out = cache
    grads = {}
    dl = dlogits * cfg.x_logit_scale
    d2 = dl.reshape(-1, dl.shape[-1])
    _acc(grads, "generator.weight", d2.T @ dec_out.reshape(-1, dec_out.shape[-1]))
    ddec = _rt(dl @ _wt(w, "generator.weight"))
    dmem = decoder_bwd(w, cfg, tgt_seq, ddec, dec_caches, grads)
    encoder_bwd(w, cfg, src_seq, dmem, enc_caches, grads)
    return grads


def tie_grads(grads, tied_proj, tied_emb):
    """lib/transformer/Models.py:259-271: aliased parameters accumulate into one .grad."""
    g = dict(grads)
    if tied_proj:
        s = g["generator.weight"] + g["decoder.tgt_word_emb.weight"]
        g["generator.weight"] = g["decoder.tgt_word_emb.weight"] = s
    if tied_emb:
        s = g["decoder.tgt_word_emb.weight"] + g["encoder.src_word_emb.weight"]
        g["decoder.tgt_word_emb.weight"] = g["encoder.src_word_emb.weight"] = s
        if tied_proj:
            g["generator.weight"] = s
    return g


# ----------------------------------------------------------------------------- loss
def ls_xent(pred, gold, smoothing, eps=0.1):
    """lib/TransformerModel.py:340-375.  pred (T,V), gold (T,) -> (loss SUM over non-pad rows,
    dpred, n_correct).  Smoothing: q = (1-eps) on gold, eps/(V-1) elsewhere."""
    T, V = pred.shape
    gold = gold.reshape(-1)
    m = pred.max(1, keepdims=True)
    lse = m + np.log(np.exp(pred - m).sum(1, keepdims=True))
    logp = pred - lse
    non_pad = gold != PAD
    if smoothing:
        q = np.full((T, V), eps / (V - 1), dtype=pred.dtype)
        q[np.arange(T), gold] = 1 - eps
    else:
        q = np.zeros((T, V), dtype=pred.dtype)
        q[np.arange(T), gold] = 1
    row = -(q * logp).sum(1)
    loss = row[non_pad].sum()
    dpred = (np.exp(logp) * q.sum(1, keepdims=True) - q) * non_pad[:, None]
    n_correct = int(((pred.argmax(1) == gold) & non_pad).sum())
    return loss, _rt(dpred.astype(pred.dtype)), n_correct  # a bf16 path stores the gradient in the logits' format


# ----------------------------------------------------------------------------- optimiser
def noam_lr(step, d_model, n_warmup):
    """lib/transformer/Optim.py:24-36 — `step` is the value AFTER the increment."""
    return np.power(d_model, -0.5) * min(np.power(step, -0.5), np.power(n_warmup, -1.5) * step)


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.98, eps=1e-9):
    """torch.optim.Adam as configured at lib/TransformerModel.py:159-165 (no weight decay,
    no amsgrad).  Arrays are updated in place; `step` starts at 1."""
    m *= b1
    m += (1 - b1) * g
    v *= b2
    v += (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p -= (lr / bc1) * (m / denom)


# ----------------------------------------------------------------------------- decode
def log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - (m + np.log(np.exp(x - m).sum(-1, keepdims=True)))


def stable_topk(flat, k):
    """Descending top-k; ties broken towards the lower index."""
    idx = np.argsort(-flat, kind="stable")[:k]
    return flat[idx], idx


class Beam:
    """lib/transformer/Beam.py:14-110 (with floor division at :70)."""

    def __init__(self, size):
        self.size, self.done = size, False
        self.scores = np.zeros(size, dtype=np.float32)
        self.prev_ks = []
        first = np.full(size, PAD, dtype=np.int64)
        first[0] = SOS
        self.next_ys = [first]

    def advance(self, word_prob):
        V = word_prob.shape[1]
        lk = word_prob + self.scores[:, None] if self.prev_ks else word_prob[0]
        best, ids = stable_topk(lk.reshape(-1), self.size)
        self.scores = best.astype(np.float32)
        prev_k = ids // V
        self.prev_ks.append(prev_k)
        self.next_ys.append(ids - prev_k * V)
        if self.next_ys[-1][0] == EOS:
            self.done = True
        return self.done

    def hypothesis(self, k):
        hyp = []
        for j in range(len(self.prev_ks) - 1, -1, -1):
            hyp.append(int(self.next_ys[j + 1][k]))
            k = self.prev_ks[j][k]
        return hyp[::-1]

    def current_state(self):
        if len(self.next_ys) == 1:
            return self.next_ys[0][:, None]
        keys = np.argsort(-self.scores, kind="stable")
        return np.array([[SOS] + self.hypothesis(k) for k in keys], dtype=np.int64)


def translate_batch(w, cfg, src_seq, src_pos, beam_size, n_best, max_token_seq_len):
    """lib/transformer/Translator.py:55-226, restated without the active-set compaction
    (finished sentences are simply skipped; results are identical because sentences do not
    interact).  The decoder re-runs on the whole prefix each step, like the reference, and
    the generator is applied WITHOUT x_logit_scale (Translator.py:137)."""
    enc_out, _, _ = encoder_fwd(w, cfg, src_seq, src_pos)
    n_inst = src_seq.shape[0]
    beams = [Beam(beam_size) for _ in range(n_inst)]
    for t in range(1, max_token_seq_len + 1):
        active = [i for i in range(n_inst) if not beams[i].done]
        if not active:
            break
        dec_seq = np.concatenate([beams[i].current_state() for i in active], 0)
        dec_pos = np.tile(np.arange(1, t + 1, dtype=np.int64)[None], (dec_seq.shape[0], 1))
        rep = np.repeat(np.array(active), beam_size)
        with np.errstate(invalid="ignore"):
            dec_out, _, _, _ = decoder_fwd(w, cfg, dec_seq, dec_pos, src_seq[rep], enc_out[rep])
        logp = log_softmax(dec_out[:, -1, :] @ w["generator.weight"].T)
        logp = logp.reshape(len(active), beam_size, -1)
        for j, i in enumerate(active):
            beams[i].advance(logp[j])
    hyps, scores = [], []
    for b in beams:
        order = np.argsort(-b.scores, kind="stable")[:n_best]
        scores.append(b.scores[order])
        hyps.append([b.hypothesis(k) for k in order])
    return hyps, scores
