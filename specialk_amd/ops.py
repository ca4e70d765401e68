"""torch.autograd glue over the HIP kernels.

Each Function's forward/backward enqueues C-ABI calls (specialk_amd._lib) on the current
stream; no arithmetic on the hot path is done by torch itself.  Parameters stay fp32
("master") nn.Parameters so that state_dicts match the reference; when the compute dtype is
bfloat16 the kernels read bf16 working copies kept by `compute_weight` and weight
gradients are produced directly in fp32.
"""
import itertools
import weakref

import torch

from . import _lib as L

PAD = 0

# ----------------------------------------------------------------------------- dropout seeds
_seed_base = 0x5EED5EED
_seed_counter = itertools.count(1)


def manual_seed(seed):
    """Reseed the kernels' counter-based dropout generator (torch.manual_seed does not)."""
    global _seed_base, _seed_counter
    _seed_base = int(seed) & 0xFFFFFFFFFFFF
    _seed_counter = itertools.count(1)


def next_seed():
    return (_seed_base * 1000003 + next(_seed_counter) * 0x9E3779B1) & 0x7FFFFFFFFFFFFFFF


# ----------------------------------------------------------------------------- compute copies
class _Copy:
    __slots__ = ("ref", "version", "ptr", "tensor")


_copies = {}


def compute_weight(p, dtype):
    """The tensor the kernels read for parameter `p`: p itself for fp32 compute, else a cached
    bf16 copy refreshed whenever p was modified through torch (version counter) or re-allocated.
    The fused Adam kernel updates the copy in place, so no refresh happens on the hot path."""
    t = p.detach()
    if dtype == torch.float32:
        return t
    key = id(p)
    c = _copies.get(key)
    if c is None or c.ref() is not p or c.version != p._version or c.ptr != p.data_ptr():
        if c is None or c.ref() is not p or c.tensor.shape != p.shape or c.tensor.device != p.device:
            c = _Copy()
            c.ref = weakref.ref(p, lambda _r, k=key: _copies.pop(k, None))
            c.tensor = torch.empty(p.shape, dtype=dtype, device=p.device)
            _copies[key] = c
        L.convert(t.contiguous(), c.tensor)
        c.version, c.ptr = p._version, p.data_ptr()
    return c.tensor


def registered_copy(p):
    c = _copies.get(id(p))
    return c.tensor if c is not None and c.ref() is p else None


def _bias_grad(dy, T, N):
    """Column sums of dy, unless the LayerNorm backward that produced dy already attached them."""
    pre = getattr(dy, "_sk_colsum", None)
    if pre is not None and pre[0] == dy.data_ptr():
        return pre[1]
    return L.colsum(dy, T, N, N)


def _w2d(w):
    return w.reshape(w.shape[0], -1)  # Conv1d(k=1) weights (out,in,1) are linear maps (out,in)


# ----------------------------------------------------------------------------- linear
class LinearFn(torch.autograd.Function):
    """y = alpha * x W^T + b   (nn.Linear; lib/transformer/SubLayers.py:46-48,62, Models.py:278)."""

    @staticmethod
    def forward(ctx, x, W, b, alpha):
        T, K = x.shape
        Wc = _w2d(compute_weight(W, x.dtype))
        N = Wc.shape[0]
        y = L.gemm(L.GEMM_NT, x, Wc, T, N, K, x.stride(0), K, bias=None if b is None else b.detach(), alpha=alpha)
        ctx.save_for_backward(x, W)
        ctx.alpha, ctx.has_bias = alpha, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        T, K = x.shape
        Wc = _w2d(compute_weight(W, x.dtype))
        N = Wc.shape[0]
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = L.gemm(L.GEMM_NN, dy, Wc, T, K, N, N, K, alpha=ctx.alpha)
        if ctx.needs_input_grad[1]:
            dW = L.gemm(L.GEMM_TN, dy, x, N, K, T, N, x.stride(0), alpha=ctx.alpha, out_dtype=torch.float32)
            dW = dW.view(W.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = L.colsum(dy, T, N, N)
        return dx, dW, db, None


def linear(x, W, b=None, alpha=1.0):
    return LinearFn.apply(x, W, b, alpha)


# ----------------------------------------------------------------------------- position-wise FFN
class FFNFn(torch.autograd.Function):
    """w_2(relu(w_1(x)))  (lib/transformer/SubLayers.py:80-82), bias + ReLU fused into the first
    GEMM's epilogue, the ReLU backward into the input-gradient GEMM's epilogue."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        T, d = x.shape
        W1c, W2c = _w2d(compute_weight(W1, x.dtype)), _w2d(compute_weight(W2, x.dtype))
        F = W1c.shape[0]
        h = L.gemm(L.GEMM_NT, x, W1c, T, F, d, x.stride(0), d, bias=b1.detach(), epilogue=L.EPI_RELU)
        y = L.gemm(L.GEMM_NT, h, W2c, T, d, F, F, F, bias=b2.detach())
        ctx.save_for_backward(x, h, W1, W2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, W1, W2 = ctx.saved_tensors
        dy = dy.contiguous()
        T, d = x.shape
        W1c, W2c = _w2d(compute_weight(W1, x.dtype)), _w2d(compute_weight(W2, x.dtype))
        F = W1c.shape[0]
        dh = L.gemm(L.GEMM_NN, dy, W2c, T, F, d, d, F, aux=h, epilogue=L.EPI_RELU_MASK)
        dW2 = L.gemm(L.GEMM_TN, dy, h, d, F, T, d, F, out_dtype=torch.float32).view(W2.shape)
        db2 = _bias_grad(dy, T, d)
        dx = L.gemm(L.GEMM_NN, dh, W1c, T, d, F, F, d)
        dW1 = L.gemm(L.GEMM_TN, dh, x, F, d, T, F, x.stride(0), out_dtype=torch.float32).view(W1.shape)
        db1 = L.colsum(dh, T, F, F)
        return dx, dW1, db1, dW2, db2


# ----------------------------------------------------------------------------- multi-head attention
class MHAFn(torch.autograd.Function):
    """Projections + scaled-dot-product attention + output projection
    (lib/transformer/SubLayers.py:46-62 without the LayerNorm, lib/transformer/Modules.py:17-28).

    q_in (B*Lq, d), kv_in (B*Lk, d) — or, with cfg["q_off"] / cfg["k_off"] (int32 row offsets per
    sentence), packed token rows without padding.  Q/K/V projections land in column slices of one buffer and
    the attention kernel reads heads in place (no permute/contiguous copies); masks come from
    key_pad (B,Lk) uint8 / causal / an optional dense (B,Lq,Lk) uint8 mask.
    Returns (fc_out (B*Lq, d), attn (h*B, Lq, Lk) or None)."""

    @staticmethod
    def forward(ctx, q_in, kv_in, Wq, bq, Wk, bk, Wv, bv, Wo, bo, cfg):
        B, Lq, Lk, H, dk, dv = cfg["B"], cfg["Lq"], cfg["Lk"], cfg["H"], cfg["dk"], cfg["dv"]
        dt, dev = q_in.dtype, q_in.device
        d = q_in.shape[1]
        Tq, Tk = q_in.shape[0], kv_in.shape[0]  # B*Lq, B*Lk padded; token counts when packed
        self_attn = cfg["self_attn"]
        Wqc, Wkc, Wvc, Woc = (compute_weight(w, dt) for w in (Wq, Wk, Wv, Wo))
        nq, nk, nv = H * dk, H * dk, H * dv
        if self_attn:
            qkv = torch.empty((Tq, nq + nk + nv), dtype=dt, device=dev)
            ld = qkv.stride(0)
            Q, K, V = qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:]
            ldq = ldk = ldv = ld
        else:
            Q = torch.empty((Tq, nq), dtype=dt, device=dev)
            kv = torch.empty((Tk, nk + nv), dtype=dt, device=dev)
            K, V = kv[:, :nk], kv[:, nk:]
            ldq, ldk, ldv = nq, nk + nv, nk + nv
        L.gemm(L.GEMM_NT, q_in, Wqc, Tq, nq, d, q_in.stride(0), d, out=Q, ldc=ldq, bias=bq.detach())
        L.gemm(L.GEMM_NT, kv_in, Wkc, Tk, nk, d, kv_in.stride(0), d, out=K, ldc=ldk, bias=bk.detach())
        L.gemm(L.GEMM_NT, kv_in, Wvc, Tk, nv, d, kv_in.stride(0), d, out=V, ldc=ldv, bias=bv.detach())
        O = torch.empty((Tq, nv), dtype=dt, device=dev)
        lse = torch.empty((B, H, Lq), dtype=torch.float32, device=dev)
        P = torch.empty((H * B, Lq, Lk), dtype=dt, device=dev) if cfg["want_attn"] else None
        seed = next_seed() if cfg["p_attn"] > 0 else 0
        L.attn_fwd(Q, K, V, O, lse, B, H, Lq, Lk, dk, dv, ldq, ldk, ldv, nv, cfg["scale"], key_pad=cfg["key_pad"],
                   dense_mask=cfg["dense_mask"], causal=cfg["causal"], p_drop=cfg["p_attn"], seed=seed, P_out=P,
                   q_off=cfg.get("q_off"), k_off=cfg.get("k_off"))
        out = L.gemm(L.GEMM_NT, O, Woc, Tq, d, nv, nv, nv, bias=bo.detach())
        ctx.save_for_backward(q_in, kv_in, Q, K, V, O, lse, Wq, Wk, Wv, Wo)
        ctx.cfg, ctx.seed, ctx.lds = cfg, seed, (ldq, ldk, ldv)
        if P is not None:
            ctx.mark_non_differentiable(P)
        return out, P

    @staticmethod
    def backward(ctx, dout, _dP):
        q_in, kv_in, Q, K, V, O, lse, Wq, Wk, Wv, Wo = ctx.saved_tensors
        cfg = ctx.cfg
        B, Lq, Lk, H, dk, dv = cfg["B"], cfg["Lq"], cfg["Lk"], cfg["H"], cfg["dk"], cfg["dv"]
        dt, dev = q_in.dtype, q_in.device
        d = q_in.shape[1]
        Tq, Tk = q_in.shape[0], kv_in.shape[0]
        nq, nk, nv = H * dk, H * dk, H * dv
        ldq, ldk, ldv = ctx.lds
        dout = dout.contiguous()
        Wqc, Wkc, Wvc, Woc = (compute_weight(w, dt) for w in (Wq, Wk, Wv, Wo))
        f32 = torch.float32
        # output projection
        dO = L.gemm(L.GEMM_NN, dout, Woc, Tq, nv, d, d, nv)
        dWo = L.gemm(L.GEMM_TN, dout, O, d, nv, Tq, d, nv, out_dtype=f32)
        dbo = _bias_grad(dout, Tq, d)
        # attention core
        if cfg["self_attn"]:
            dqkv = torch.empty((Tq, nq + nk + nv), dtype=dt, device=dev)
            dQ, dK, dV = dqkv[:, :nq], dqkv[:, nq:nq + nk], dqkv[:, nq + nk:]
            lddq = lddk = lddv = dqkv.stride(0)
        else:
            dQ = torch.empty((Tq, nq), dtype=dt, device=dev)
            dkv = torch.empty((Tk, nk + nv), dtype=dt, device=dev)
            dK, dV = dkv[:, :nk], dkv[:, nk:]
            lddq, lddk, lddv = nq, nk + nv, nk + nv
        delta = torch.empty((B, H, Lq), dtype=f32, device=dev)
        L.attn_bwd(Q, K, V, O, dO, lse, dQ, dK, dV, delta, B, H, Lq, Lk, dk, dv, ldq, ldk, ldv, nv, lddq, lddk, lddv,
                   cfg["scale"], key_pad=cfg["key_pad"], dense_mask=cfg["dense_mask"], causal=cfg["causal"],
                   p_drop=cfg["p_attn"], seed=ctx.seed, q_off=cfg.get("q_off"), k_off=cfg.get("k_off"))
        # projections
        dWq = L.gemm(L.GEMM_TN, dQ, q_in, nq, d, Tq, lddq, q_in.stride(0), out_dtype=f32)
        dWk = L.gemm(L.GEMM_TN, dK, kv_in, nk, d, Tk, lddk, kv_in.stride(0), out_dtype=f32)
        dWv = L.gemm(L.GEMM_TN, dV, kv_in, nv, d, Tk, lddv, kv_in.stride(0), out_dtype=f32)
        dbq, dbk, dbv = L.colsum(dQ, Tq, nq, lddq), L.colsum(dK, Tk, nk, lddk), L.colsum(dV, Tk, nv, lddv)
        dq_in = L.gemm(L.GEMM_NN, dQ, Wqc, Tq, d, nq, lddq, d)
        if cfg["self_attn"]:
            L.gemm(L.GEMM_NN, dK, Wkc, Tk, d, nk, lddk, d, out=dq_in, ldc=d, epilogue=L.EPI_ACCUM)
            L.gemm(L.GEMM_NN, dV, Wvc, Tk, d, nv, lddv, d, out=dq_in, ldc=d, epilogue=L.EPI_ACCUM)
            dkv_in = None
        else:
            dkv_in = L.gemm(L.GEMM_NN, dK, Wkc, Tk, d, nk, lddk, d)
            L.gemm(L.GEMM_NN, dV, Wvc, Tk, d, nv, lddv, d, out=dkv_in, ldc=d, epilogue=L.EPI_ACCUM)
        return dq_in, dkv_in, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo, None


class SDPAFn(torch.autograd.Function):
    """softmax(mask(q k^T / T)) v with dropout on the probabilities, on head-major batches
    q (n,Lq,dk), k (n,Lk,dk), v (n,Lk,dv) and an optional bool mask (n,Lq,Lk) — the
    signature of ScaledDotProductAttention.forward (lib/transformer/Modules.py:17-28)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, scale, p_drop):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        n, Lq, dk = q.shape
        Lk, dv = k.shape[1], v.shape[2]
        dm = None if mask is None else mask.to(torch.uint8).contiguous()
        O = torch.empty((n, Lq, dv), dtype=q.dtype, device=q.device)
        lse = torch.empty((n, 1, Lq), dtype=torch.float32, device=q.device)
        P = torch.empty((n, Lq, Lk), dtype=q.dtype, device=q.device)
        seed = next_seed() if p_drop > 0 else 0
        L.attn_fwd(q, k, v, O, lse, n, 1, Lq, Lk, dk, dv, dk, dk, dv, dv, scale, dense_mask=dm, p_drop=p_drop,
                   seed=seed, P_out=P)
        ctx.save_for_backward(q, k, v, O, lse)
        ctx.dm, ctx.scale, ctx.p_drop, ctx.seed = dm, scale, p_drop, seed
        ctx.mark_non_differentiable(P)
        return O, P

    @staticmethod
    def backward(ctx, dO, _dP):
        q, k, v, O, lse = ctx.saved_tensors
        n, Lq, dk = q.shape
        Lk, dv = k.shape[1], v.shape[2]
        dq, dk_, dv_ = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty((n, 1, Lq), dtype=torch.float32, device=q.device)
        L.attn_bwd(q, k, v, O, dO.contiguous(), lse, dq, dk_, dv_, delta, n, 1, Lq, Lk, dk, dv, dk, dk, dv, dv, dk, dk,
                   dv, ctx.scale, dense_mask=ctx.dm, p_drop=ctx.p_drop, seed=ctx.seed)
        return dq, dk_, dv_, None, None, None


# ----------------------------------------------------------------------------- residual + LN
class AddLNFn(torch.autograd.Function):
    """LayerNorm(res + dropout(x)) with pad rows zeroed (lib/transformer/SubLayers.py:62-63,83-84
    and the `*= non_pad_mask` of lib/transformer/Layers.py)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, row_pad, eps, p_drop):
        seed = next_seed() if p_drop > 0 else 0
        x, res = x.contiguous(), res.contiguous()
        y, mean, rstd = L.add_ln_fwd(x, res, gamma.detach(), beta.detach(), row_pad, eps, p_drop, seed)
        ctx.save_for_backward(x, res, gamma, mean, rstd)
        ctx.row_pad, ctx.p_drop, ctx.seed = row_pad, p_drop, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, gamma, mean, rstd = ctx.saved_tensors
        dx, dres, dg, db, dxs = L.add_ln_bwd(dy.contiguous(), x, res, gamma.detach(), ctx.row_pad, mean, rstd,
                                             ctx.p_drop, ctx.seed, want_dx_colsum=True)
        if dx.data_ptr() == dres.data_ptr():
            dx = dx.view(dx.shape)  # distinct tensor object for the x-branch gradient
        dx._sk_colsum = (dx.data_ptr(), dxs)  # bias gradient of the linear layer that produced x
        return dx, dres, dg, db, None, None, None


def add_ln(x, res, gamma, beta, row_pad=None, eps=1e-5, p_drop=0.0):
    return AddLNFn.apply(x, res, gamma, beta, row_pad, eps, p_drop)


# ----------------------------------------------------------------------------- embeddings
class EmbedFn(torch.autograd.Function):
    """word_emb[seq] + pos_table[pos]  (lib/transformer/Models.py:117-118,181-182)."""

    @staticmethod
    def forward(ctx, seq, pos, word_emb, pos_table, dtype):
        seq, pos = seq.contiguous().view(-1), pos.contiguous().view(-1)
        out = L.embed_fwd(seq, pos, word_emb.detach(), pos_table.detach(), dtype)
        ctx.save_for_backward(seq)
        ctx.shape = word_emb.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (seq,) = ctx.saved_tensors
        dW = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        L.embed_bwd(seq, dout.contiguous(), dW, PAD)
        return None, None, dW, None, None


# ----------------------------------------------------------------------------- loss
class XentFn(torch.autograd.Function):
    """Label-smoothed / plain cross entropy, summed over non-pad rows
    (lib/TransformerModel.py:353-375).  One fused pass yields loss, accuracy counts and the
    gradient; backward only scales the stored gradient."""

    @staticmethod
    def forward(ctx, logits, gold, eps, logit_scale, stats, unit_grad):
        logits = logits.contiguous()
        ctx.unit_grad = unit_grad
        want_grad = ctx.needs_input_grad[0]
        dlogits = torch.empty_like(logits) if want_grad else None
        L.xent_fwd_bwd(logits, gold.contiguous().view(-1), dlogits, stats, logit_scale, eps, PAD)
        ctx.dlogits = dlogits
        return stats[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        d = ctx.dlogits
        ctx.dlogits = None
        if not ctx.unit_grad:
            d = d * g.to(d.dtype)
        return d, None, None, None, None, None


def xent(logits, gold, smoothing, logit_scale=1.0, unit_grad=False):
    """Returns (loss, stats) with stats = float64 [loss_sum, n_correct, n_non_pad] on device.
    unit_grad=True promises that the loss is differentiated directly (loss.backward()), which
    lets backward hand out the stored gradient without a scaling pass over (T,V)."""
    stats = torch.zeros(3, dtype=torch.float64, device=logits.device)
    loss = XentFn.apply(logits, gold, 0.1 if smoothing else 0.0, logit_scale, stats, unit_grad)
    return loss, stats
