"""torch.autograd glue over the HIP kernels.

Each Function's forward/backward enqueues C-ABI calls (specialk_amd._lib) on the current
stream; no arithmetic on the hot path is done by torch itself.  Parameters stay fp32
("master") nn.Parameters so that state_dicts match the reference; when the compute dtype is
bfloat16 the kernels read bf16 working copies kept by `compute_weight` and weight
gradients are produced directly in fp32.
"""
import itertools
import weakref

import contextlib
import os

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


# ----------------------------------------------------------------------------- derived copies
class _Derived:
    __slots__ = ("refs", "versions", "ptrs", "tensor", "rows")


_derived = {}
_derived_tables = None  # device tables of every registered derived copy, for the post-step refresh


def _as2d(p):
    return p.shape[0], p.numel() // p.shape[0]


def derived_weight(kind, params):
    """Working copies derived from fp32 master parameters, kept current by `refresh_derived`
    (one launch after the optimiser step) and rebuilt lazily when a parameter was modified
    through torch:
      "cat"  : bf16 [sum R_i][C]   the matrices stacked by rows (one fused projection GEMM),
      "catT" : bf16 [C][sum R_i]   their transposes side by side (K-contiguous weight operand of
                                   the input-gradient GEMMs),
      "bias" : fp32 [sum n_i]      bias vectors concatenated,
      "fm"   : bf16, the matrices stacked by rows ([sum R_i][C]) in the fragment-major layout of the row-panel
               kernels ([N/64][K/32][4][64][8], see skhip_cast2d_multi; R_i % 64 == 0, C % 64 == 0),
      "fmT"  : the same of their transposes side by side ([C][sum R_i]: the weight operand of an input-gradient
               product)."""
    global _derived_tables
    key = (kind,) + tuple(id(p) for p in params)
    d = _derived.get(key)
    fresh = d is not None and all(r() is p and v == p._version and q == p.data_ptr()
                                  for r, v, q, p in zip(d.refs, d.versions, d.ptrs, params))
    if fresh:
        return d.tensor
    dev = params[0].device
    if kind == "bias":
        total = sum(p.numel() for p in params)
        shape, dt = (total,), torch.float32
    elif kind in ("fm", "fmT"):
        C = _as2d(params[0])[1]
        assert all(_as2d(p)[1] == C and _as2d(p)[0] % 64 == 0 for p in params) and C % 64 == 0
        total = sum(_as2d(p)[0] for p in params)
        shape, dt = ((total, C) if kind == "fm" else (C, total)), torch.bfloat16
    else:
        C = _as2d(params[0])[1]
        assert all(_as2d(p)[1] == C for p in params)
        total = sum(_as2d(p)[0] for p in params)
        shape, dt = ((total, C) if kind == "cat" else (C, total)), torch.bfloat16
    if d is None or d.tensor.shape != shape or d.tensor.device != dev:
        d = _Derived()
        d.tensor = torch.empty(shape, dtype=dt, device=dev)
        _derived[key] = d
        _derived_tables = None
    rows, off = [], 0
    esz = d.tensor.element_size()
    for p in params:
        if kind == "bias":
            rows.append((p.data_ptr(), d.tensor.data_ptr() + off * esz, 1, p.numel(), total, 2))
            off += p.numel()
        elif kind == "fm":  # the matrices stacked by rows: each starts a new 64-row group
            R, C = _as2d(p)
            rows.append((p.data_ptr(), d.tensor.data_ptr() + off * C * esz, R, C, 0, 2 << 2))
            off += R
        elif kind == "fmT":  # their transposes side by side: each is a run of K-steps of every group
            R, C = _as2d(p)
            rows.append((p.data_ptr(), d.tensor.data_ptr() + off * 64 * esz, R, C, total, (2 << 2) | 1))
            off += R
        elif kind == "cat":
            R, C = _as2d(p)
            rows.append((p.data_ptr(), d.tensor.data_ptr() + off * C * esz, R, C, C, 0))
            off += R
        else:
            R, C = _as2d(p)
            rows.append((p.data_ptr(), d.tensor.data_ptr() + off * esz, R, C, total, 1))
            off += R
    d.rows = rows
    d.refs = [weakref.ref(p, lambda _r, k=key: _drop_derived(k)) for p in params]
    d.versions = [p._version for p in params]
    d.ptrs = [p.data_ptr() for p in params]
    if any(q != r[0] for q, r in zip(d.ptrs, rows)):
        _derived_tables = None
    _launch_cast2d(_tables_of(rows, dev))
    _derived_tables = None  # pointers may have changed
    return d.tensor


def _drop_derived(key):
    """A master parameter of the copy `key` was freed: forget the copy and the device tables that still
    hold its (now dangling) source pointers, so the next refresh neither reads freed memory nor keeps the
    orphaned copy alive."""
    global _derived_tables
    _derived.pop(key, None)
    _derived_tables = None


def _tables_of(rows, dev):
    t = torch.tensor(rows, dtype=torch.int64)
    tiles = int((((t[:, 2] + 31) // 32) * ((t[:, 3] + 31) // 32)).max())
    return (t[:, 0].contiguous().to(dev), t[:, 1].contiguous().to(dev), t[:, 2:].contiguous().to(dev), len(rows), tiles)


def _launch_cast2d(tab):
    L.cast2d_multi(tab[0], tab[1], tab[2], tab[3], tab[4])


_weights_epoch = 0


def weights_epoch():
    """Counts the optimiser updates of this process (refresh_derived calls).  Together with the parameters' own version
    counters — which every other way of changing a parameter advances — it tells a holder of weights in some prepared
    form (the beam decoder's fragment-major matrices) whether they are still current."""
    return _weights_epoch


def refresh_derived():
    """Rewrite every derived copy from the master parameters (called by the optimiser after its
    update, whose kernel changes the parameters without touching torch's version counters)."""
    global _derived_tables, _weights_epoch
    _weights_epoch += 1
    _pending_dres.clear()   # an optimiser update ends the step: nothing of its backward pass is still on its way
    if not _derived:
        return
    if _derived_tables is None:
        dead = [k for k, d in _derived.items() if any(r() is None for r in d.refs)]
        for k in dead:
            del _derived[k]
        by_dev = {}
        for d in _derived.values():
            by_dev.setdefault(d.tensor.device, []).extend(d.rows)
        _derived_tables = [_tables_of(rows, dev) for dev, rows in by_dev.items()]
    for tab in _derived_tables:
        _launch_cast2d(tab)


def _bias_grad(dy, T, N):
    """Column sums of dy, unless the LayerNorm backward that produced dy already attached them."""
    pre = getattr(dy, "_sk_colsum", None)
    if pre is not None and pre[0] == dy.data_ptr():
        del dy._sk_colsum  # sole owner from here on: autograd can adopt the tensor as .grad instead of cloning it
        return pre[1]
    return L.colsum(dy, T, N, N)


# data_ptr of a gradient whose residual part AddLNFn.backward handed on -> not yet picked up.  Process-wide, so it is
# emptied wherever a step begins or ends (batched_weight_grads entry and exit, the optimiser's refresh_derived): a
# backward pass that was aborted between a hand-over and its pick-up must not fail the next, unrelated one.
_pending_dres = {}


def _residual_grad(dy):
    """The residual-branch gradient that AddLNFn.backward attached to dy (see there), or None."""
    pre = getattr(dy, "_sk_dres", None)
    if pre is not None and pre[0] == dy.data_ptr():
        _pending_dres.pop(pre[0], None)
        return pre[1]
    return None


def _check_residual_handoff():
    """AddLNFn.backward hands the residual-branch gradient to the backward of the sub-layer that runs next by
    attaching it to the gradient tensor.  If that tensor object was replaced on the way (a copy, a hook), the
    sub-layer would not see it and the gradient would silently lose a term: fail instead."""
    if _pending_dres:
        _pending_dres.clear()
        raise RuntimeError("a residual-branch gradient handed over by AddLNFn.backward was not picked up by the "
                           "following sub-layer's backward (the gradient tensor was replaced on the way); "
                           "call add_ln(..., res_feeds_x=False) for this graph")


def _w2d(w):
    return w.reshape(w.shape[0], -1)  # Conv1d(k=1) weights (out,in,1) are linear maps (out,in)


_ROWPANEL = os.environ.get("SK_ROWPANEL", "1") != "0"
_cu_count = {}


def _panel_rows_ok(x, T):
    """Whether T token rows suit the row-panel kernels (one workgroup per 104 rows, one per CU at a time): the
    workgroups must fill at least 80 % of the rounds of the chip they take — 25.6 k rows are 247 of 256 CUs, 33 k rows
    would be one full round and a quarter of a second one, where the tiled kernels do better."""
    if not _ROWPANEL or x.dtype != torch.bfloat16 or x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        return False
    cus = _cu_count.get(x.device)
    if cus is None:
        cus = _cu_count[x.device] = torch.cuda.get_device_properties(x.device).multi_processor_count
    cus = max(1, cus - L.reserved_cus)   # (CUs held by collectives: a round of 247 panels on 240 CUs is two rounds)
    blocks = -(-T // L.ROWPANEL_ROWS)
    rounds = -(-blocks // cus)
    return blocks >= 0.8 * rounds * cus


def _panel_weights_ok(Ws, d_in):
    """Stacked weights (n_i, 512) whose total row count is a multiple of 512: the operand shape of the panel kernel."""
    return d_in == 512 and all(_as2d(w)[1] == 512 and _as2d(w)[0] % 64 == 0 for w in Ws) and sum(_as2d(w)[0] for w in Ws) % 512 == 0


def _input_grad(dy, ldy, Ws, T, d_in, dt, alpha=1.0, aux=None, epilogue=0, add_to=None):
    """dx (T, d_in) = alpha * dy (T, sum n_i; row stride ldy) @ [W_1; W_2; ...] for weights W_i (n_i, d_in).
    bf16: ONE GEMM against the K-contiguous transposed copy; fp32: one exact-fp32 GEMM per weight.
    add_to: a (T, d_in) tensor the result is accumulated into (and which is returned)."""
    n_tot = sum(_as2d(w)[0] for w in Ws)
    if dt == torch.bfloat16 and len(Ws) == 1 and n_tot >= 8192:
        # long contraction (the generator: n = vocabulary): NN against the weight's own bf16 copy is the
        # faster form there, and no transposed copy has to be kept up to date
        return L.gemm(L.GEMM_NN, dy, _w2d(compute_weight(Ws[0], dt)), T, d_in, n_tot, ldy, d_in, alpha=alpha, aux=aux,
                      epilogue=epilogue | (L.EPI_ACCUM if add_to is not None else 0), out=add_to,
                      ldc=None if add_to is None else d_in)
    if dt == torch.bfloat16:
        return L.gemm(L.GEMM_NT, dy, derived_weight("catT", Ws), T, d_in, n_tot, ldy, n_tot, alpha=alpha, aux=aux,
                      epilogue=epilogue | (L.EPI_ACCUM if add_to is not None else 0), out=add_to,
                      ldc=None if add_to is None else d_in)
    assert len(Ws) == 1 or (aux is None and epilogue == 0)
    assert add_to is None or (aux is None and epilogue == 0)
    dx, off = add_to, 0
    for w in Ws:
        n = _as2d(w)[0]
        if dx is None:
            dx = L.gemm(L.GEMM_NN, dy[:, off:off + n], _w2d(w.detach()), T, d_in, n, ldy, d_in, alpha=alpha, aux=aux,
                        epilogue=epilogue)
        else:
            L.gemm(L.GEMM_NN, dy[:, off:off + n], _w2d(w.detach()), T, d_in, n, ldy, d_in, alpha=alpha, out=dx, ldc=d_in,
                   epilogue=L.EPI_ACCUM)
        off += n
    return dx


def _project(x, Ws, bs, out, ldc, dt):
    """out[:, blocks] = x @ W_i^T + b_i for the stacked weights: one fused GEMM in bf16, one GEMM each in fp32."""
    T, d = x.shape
    n_tot = sum(_as2d(w)[0] for w in Ws)
    if dt == torch.bfloat16 and n_tot >= 1024 and _panel_weights_ok(Ws, d) and _panel_rows_ok(x, T) and ldc % 8 == 0 \
            and out.data_ptr() % 16 == 0:
        # several column blocks of 512 against the same rows: the rows are brought on chip once
        b = derived_weight("bias", bs) if len(bs) > 1 else bs[0].detach()
        L.rowpanel_gemm(x, derived_weight("fm", Ws), n_tot, bias=b, out=out, ldc=ldc)
        return
    if dt == torch.bfloat16:
        W = derived_weight("cat", Ws) if len(Ws) > 1 else _w2d(compute_weight(Ws[0], dt))
        b = derived_weight("bias", bs) if len(bs) > 1 else bs[0].detach()
        L.gemm(L.GEMM_NT, x, W, T, n_tot, d, x.stride(0), d, out=out, ldc=ldc, bias=b)
        return
    off = 0
    for w, b in zip(Ws, bs):
        n = _as2d(w)[0]
        L.gemm(L.GEMM_NT, x, _w2d(w.detach()), T, n, d, x.stride(0), d, out=out[:, off:off + n], ldc=ldc, bias=b.detach())
        off += n


# ----------------------------------------------------------------------------- batched weight gradients
class _WeightGradBatch:
    def __init__(self, on_ready, flush_tiles):
        self.on_ready, self.flush_tiles = on_ready, flush_tiles
        self.items, self.tiles, self.seen = [], 0, set()


_wgrad_batch = None


@contextlib.contextmanager
def batched_weight_grads(on_ready=None, flush_tiles=256):
    """Wrap `loss.backward()` in this context to compute the bf16 weight gradients of the attention and
    feed-forward sub-layers in grouped launches instead of one split-K GEMM each.

    Inside it, MHAFn / FFNFn hand autograd no gradient for their weight matrices; the products
    dW = dy^T x are queued (their operands stay alive) and run together — whenever at least
    `flush_tiles` 256 x 256 output tiles are waiting (one round of the chip) and when the context
    exits — writing straight into `W.grad` (created if absent, accumulated into otherwise, so flat
    gradient buckets keep working).  A (512, 512) weight is four tiles: alone it has to be cut into
    dozens of K splits with their slab traffic to occupy the chip, in a group it is four of several
    hundred equal work items.  (Thresholds of 512 or 1024 tiles measured 3 % slower end to end than 256:
    the queued operands leave the caches before they are used.)  on_ready(params) is called after each grouped launch with the
    parameters whose gradients it wrote (a data-parallel reducer's cue; autograd hooks do not fire
    for these parameters).  The bias gradient that goes with a queued weight gradient (the column sums of the same
    dy) is computed by the same launch on the matrix pipe and written to `b.grad` alike — straight into the
    reducer's bucket when on_ready belongs to an object with a bucket_view(p) method.  Output-projection bias,
    LayerNorm and embedding gradients still flow through autograd.  Inside the context the backward pass runs on
    the calling thread (torch.autograd.set_multithreading_enabled(False); restored on exit)."""
    global _wgrad_batch
    if _wgrad_batch is not None:
        raise RuntimeError("batched_weight_grads() does not nest")
    _wgrad_batch = _WeightGradBatch(on_ready, flush_tiles)
    _pending_dres.clear()
    # One process drives one GPU: the autograd engine's per-device worker thread only adds a hand-over per node
    # (a 2-layer model's step is host-bound: 3.16 -> 2.73 ms with the backward pass run on the calling thread).
    mt = torch.autograd.is_multithreading_enabled()
    torch.autograd.set_multithreading_enabled(False)
    try:
        yield
        _flush_weight_grads()
    finally:
        torch.autograd.set_multithreading_enabled(mt)
        _wgrad_batch = None
        _pending_dres.clear()


def settle_host_garbage():
    """Collect the interpreter's garbage now and move everything that survives into the permanent generation
    (gc.freeze): a full collection walks every object torch and the model have created — 100 ms here, which lands in
    the middle of some training step every few dozen steps — while after the freeze it only walks what was created
    since.  Called by the trainer at the start of an epoch and by bench.py after its warm-up steps."""
    import gc
    gc.collect()
    gc.freeze()


def _reducer():
    """The data-parallel reducer whose mark_ready the current batched_weight_grads() context reports to, if it
    hands out its bucket slices (claim), else None."""
    b = _wgrad_batch
    r = getattr(b.on_ready, "__self__", None) if b is not None and b.on_ready is not None else None
    return r if r is not None and hasattr(r, "claim") else None


def _claim_grad(p):
    """(bucket slice to write the gradient of p into, accumulate?) inside a batched_weight_grads() context that
    reports to a reducer; (None, False) otherwise.  For producers whose result reaches .grad through autograd: the
    first producer of the step overwrites the slice and returns a fresh alias of it (autograd adopts it, so .grad
    IS the slice); a later producer of the same matrix adds to the slice in place and returns None."""
    r = _reducer()
    return r.claim(p) if r is not None and p.grad is None else (None, False)


def _grad_target(p, fixups, reducer=None):
    """(tensor the launch writes, accumulate?) for the gradient of p: p.grad itself when it can be written in
    place, the reducer's bucket slice or a fresh tensor that becomes p.grad when there is none, else a temporary
    that is added afterwards."""
    late = reducer.late_target(p) if reducer is not None and hasattr(reducer, "late_target") else None
    if late is not None:  # the bucket that holds this gradient has already started its all-reduce
        return late, False
    g = p.grad
    if g is None:
        view, acc = reducer.claim(p) if reducer is not None else (None, False)
        if view is not None:
            if not acc:  # (acc: an alias of the slice is already on its way to .grad through autograd)
                p.grad = view
            return view, acc
        g = p.grad = torch.empty(p.shape, dtype=torch.float32, device=p.device)
        return g, False
    if g.dtype == torch.float32 and g.is_contiguous():
        return g, True
    out = torch.empty(p.shape, dtype=torch.float32, device=p.device)
    fixups.append((g, out))
    return out, False


def _flush_weight_grads():
    b = _wgrad_batch
    if b is None or not b.items:
        return
    items, b.items, b.tiles, b.seen = b.items, [], 0, set()
    problems, fixups, params = [], [], []
    reducer = _reducer()
    with torch.no_grad():
        for dy, x, W, n, d_in, T, bias in items:
            out, acc = _grad_target(W, fixups, reducer)
            cs, cs_acc = (None, False) if bias is None else _grad_target(bias, fixups, reducer)
            problems.append((dy, x, out, n, d_in, T, dy.stride(0), x.stride(0), d_in, acc, cs, cs_acc))
            params.append(W)
            if bias is not None:
                params.append(bias)
        L.gemm_grouped_tn(problems)
        for g, out in fixups:
            g.add_(out)
    if b.on_ready is not None:
        b.on_ready(params)


def _batchable(dy, x):
    n, d_in = dy.shape[1], x.shape[1]
    return not (_wgrad_batch is None or dy.dtype != torch.bfloat16 or dy.stride(0) % 8 or x.stride(0) % 8 or n % 8 or d_in % 8
                or dy.data_ptr() % 16 or x.data_ptr() % 16)


def _weight_grad(dy, x, W, T, needed=True, bias=None):
    """dW (n, d_in) = dy (T, n; any row stride)^T x (T, d_in) for the weight W of shape (n, d_in[, 1]):
    returned for autograd, or — inside batched_weight_grads(), bf16 only — queued for the next grouped
    launch, in which case None is returned and W.grad is written by that launch.
    bias: the layer's bias when its gradient (the column sums of dy) is wanted as well.  Returns (dW, db);
    queued: (None, None), and the launch writes bias.grad too."""
    n, d_in = dy.shape[1], x.shape[1]
    b = _wgrad_batch
    if not needed or not _batchable(dy, x):
        dW = None
        if needed:
            dW = L.gemm(L.GEMM_TN, dy, x, n, d_in, T, dy.stride(0), x.stride(0), out_dtype=torch.float32).view(W.shape)
        return dW, (None if bias is None else L.colsum(dy, T, n, dy.stride(0)))
    r = _reducer()
    if r is not None and hasattr(r, "direct"):  # counted when the grouped launch reports them, not by autograd's hooks
        r.direct(W)
        if bias is not None:
            r.direct(bias)
    if W.data_ptr() in b.seen:  # two products into one gradient must not share a launch
        _flush_weight_grads()
    b.seen.add(W.data_ptr())
    b.items.append((dy, x, W, n, d_in, T, bias))
    b.tiles += ((n + 255) // 256) * ((d_in + 255) // 256)
    if b.tiles >= b.flush_tiles:
        _flush_weight_grads()
    return None, None


def _unalias(dres, dy):
    """The residual-branch gradient is accumulated into in place; with dropout off it shares dy's
    storage, which queued weight-gradient products still have to read."""
    if dres is not None and _wgrad_batch is not None and dres.data_ptr() == dy.data_ptr():
        return dres.clone()
    return dres


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
        N = _as2d(W)[0]
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dy, N, (W,), T, K, x.dtype, alpha=ctx.alpha)
        if ctx.needs_input_grad[1]:
            view, acc = _claim_grad(W)
            if view is not None:
                L.gemm(L.GEMM_TN, dy, x, N, K, T, N, x.stride(0), out=view.view(N, K), ldc=K, alpha=ctx.alpha,
                       epilogue=L.EPI_ACCUM if acc else 0)
                dW = None if acc else view.view(W.shape)
            else:
                dW = L.gemm(L.GEMM_TN, dy, x, N, K, T, N, x.stride(0), alpha=ctx.alpha, out_dtype=torch.float32)
                dW = dW.view(W.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = L.colsum(dy, T, N, N)
        return dx, dW, db, None


def linear(x, W, b=None, alpha=1.0):
    return call(LinearFn, x, W, b, alpha)


# ----------------------------------------------------------------------------- position-wise FFN
class FFNFn(torch.autograd.Function):
    """w_2(relu(w_1(x)))  (lib/transformer/SubLayers.py:80-82), bias + ReLU fused into the first
    GEMM's epilogue, the ReLU backward into the input-gradient GEMM's epilogue."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        T, d = x.shape
        W2c = _w2d(compute_weight(W2, x.dtype))
        F = _as2d(W1)[0]
        # bf16: the first GEMM also writes the sign pattern of h, one bit per element, which is all the
        # backward of the ReLU needs (the input-gradient GEMM then reads 1/16 of the bytes of h)
        signs, panel = None, False
        if x.dtype == torch.bfloat16 and F % 64 == 0 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0:
            panel = _panel_weights_ok((W1,), d) and _panel_rows_ok(x, T)
            # (the row-panel kernels keep the pattern in a layout of their own, which only they read back)
            signs = L.rowpanel_pattern(T, F, x.device) if panel else torch.empty((T, F // 16), dtype=torch.int16, device=x.device)
        if panel:
            h = L.rowpanel_gemm(x, derived_weight("fm", (W1,)), F, bias=b1.detach(), signs=signs,
                                epilogue=L.EPI_RELU | L.EPI_RELU_BITS_OUT)
        else:
            h = L.gemm(L.GEMM_NT, x, _w2d(compute_weight(W1, x.dtype)), T, F, d, x.stride(0), d, bias=b1.detach(), aux=signs,
                       epilogue=L.EPI_RELU | (L.EPI_RELU_BITS_OUT if signs is not None else 0))
        y = L.gemm(L.GEMM_NT, h, W2c, T, d, F, F, F, bias=b2.detach())
        ctx.save_for_backward(x, h, W1, W2, b1)
        ctx.signs, ctx.signs_panel = signs, panel
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, W1, W2, b1 = ctx.saved_tensors
        dres = _residual_grad(dy)
        dy = dy.contiguous()
        dres = _unalias(dres, dy)
        T, d = x.shape
        F = _as2d(W1)[0]
        if ctx.signs_panel and _panel_rows_ok(dy, T):
            dh = L.rowpanel_gemm(dy, derived_weight("fmT", (W2,)), F, signs=ctx.signs, epilogue=L.EPI_RELU_MASK_BITS)
        elif ctx.signs is not None and not ctx.signs_panel:
            dh = _input_grad(dy, d, (W2,), T, F, x.dtype, aux=ctx.signs, epilogue=L.EPI_RELU_MASK_BITS)
        else:
            dh = _input_grad(dy, d, (W2,), T, F, x.dtype, aux=h, epilogue=L.EPI_RELU_MASK)
        dW2, _ = _weight_grad(dy, h, W2, T, ctx.needs_input_grad[3])
        db2 = _bias_grad(dy, T, d)
        dx = _input_grad(dh, F, (W1,), T, d, x.dtype, add_to=dres)  # accumulates into dres in place
        dW1, db1 = _weight_grad(dh, x, W1, T, ctx.needs_input_grad[1], bias=b1 if ctx.needs_input_grad[2] else None)
        return dx, dW1, db1, dW2, db2


# ----------------------------------------------------------------------------- multi-head attention
class MHAFn(torch.autograd.Function):
    """Projections + scaled-dot-product attention + output projection
    (lib/transformer/SubLayers.py:46-62 without the LayerNorm, lib/transformer/Modules.py:17-28).

    q_in (B*Lq, d), kv_in (B*Lk, d) — or, with cfg["q_off"] / cfg["k_off"] (int32 row offsets per
    sentence), packed token rows without padding.  Q/K/V projections land in column slices of one
    buffer ([q|k|v] for self-attention, q and [k|v] otherwise; in bf16 each buffer is filled by ONE
    GEMM against the stacked weights) and the attention kernel reads heads in place (no
    permute/contiguous copies); masks come from key_pad (B,Lk) uint8 / causal / an optional dense
    (B,Lq,Lk) uint8 mask.  Returns (fc_out (B*Lq, d), attn (h*B, Lq, Lk) or None)."""

    @staticmethod
    def forward(ctx, q_in, kv_in, Wq, bq, Wk, bk, Wv, bv, Wo, bo, cfg):
        return _mha_forward(ctx, q_in, kv_in, Wq, bq, Wk, bk, Wv, bv, Wo, bo, cfg, None)

    @staticmethod
    def backward(ctx, dout, _dP):
        dres = _residual_grad(dout)
        dout = dout.contiguous()
        return _mha_backward(ctx, dout, _unalias(dres, dout), None)


def _mha_forward(ctx, q_in, kv_in, Wq, bq, Wk, bk, Wv, bv, Wo, bo, cfg, ln):
    """MHAFn.forward; ln = (gamma, beta): MHALNFn.forward, which ends with the output projection and the sub-layer
    tail in one launch and returns (LayerNorm output, attn)."""
    if True:
        B, Lq, Lk, H, dk, dv = cfg["B"], cfg["Lq"], cfg["Lk"], cfg["H"], cfg["dk"], cfg["dv"]
        dt, dev = q_in.dtype, q_in.device
        d = q_in.shape[1]
        Tq, Tk = q_in.shape[0], kv_in.shape[0]  # B*Lq, B*Lk padded; token counts when packed
        self_attn = cfg["self_attn"]
        nq, nk, nv = H * dk, H * dk, H * dv
        if self_attn:
            qkv = torch.empty((Tq, nq + nk + nv), dtype=dt, device=dev)
            ldq = ldk = ldv = qkv.stride(0)
            Q, K, V = qkv[:, :nq], qkv[:, nq:nq + nk], qkv[:, nq + nk:]
            _project(q_in, (Wq, Wk, Wv), (bq, bk, bv), qkv, ldq, dt)
        elif cfg.get("kv_pre") is not None:
            # kv_in holds the k | v projections of every decoder layer (CrossKVFn); this layer's are a column slice
            c0 = cfg["kv_pre"].col
            Q = torch.empty((Tq, nq), dtype=dt, device=dev)
            K, V = kv_in[:, c0:c0 + nk], kv_in[:, c0 + nk:c0 + nk + nv]
            ldq, ldk, ldv = nq, kv_in.stride(0), kv_in.stride(0)
            _project(q_in, (Wq,), (bq,), Q, ldq, dt)
        else:
            Q = torch.empty((Tq, nq), dtype=dt, device=dev)
            kv = torch.empty((Tk, nk + nv), dtype=dt, device=dev)
            K, V = kv[:, :nk], kv[:, nk:]
            ldq, ldk, ldv = nq, nk + nv, nk + nv
            _project(q_in, (Wq,), (bq,), Q, ldq, dt)
            _project(kv_in, (Wk, Wv), (bk, bv), kv, ldk, dt)
        O = torch.empty((Tq, nv), dtype=dt, device=dev)
        lse = torch.empty((B, H, Lq), dtype=torch.float32, device=dev)
        P = torch.empty((H * B, Lq, Lk), dtype=dt, device=dev) if cfg["want_attn"] else None
        seed = next_seed() if cfg["p_attn"] > 0 else 0
        L.attn_fwd(Q, K, V, O, lse, B, H, Lq, Lk, dk, dv, ldq, ldk, ldv, nv, cfg["scale"], key_pad=cfg["key_pad"],
                   dense_mask=cfg["dense_mask"], causal=cfg["causal"], p_drop=cfg["p_attn"], seed=seed, P_out=P,
                   q_off=cfg.get("q_off"), k_off=cfg.get("k_off"))
        ctx.cfg, ctx.seed, ctx.lds = cfg, seed, (ldq, ldk, ldv)
        if P is not None:
            ctx.mark_non_differentiable(P)
        if ln is None:
            out = L.gemm(L.GEMM_NT, O, compute_weight(Wo, dt), Tq, d, nv, nv, nv, bias=bo.detach())
            ctx.save_for_backward(q_in, kv_in, Q, K, V, O, lse, Wq, Wk, Wv, Wo, bq, bk, bv)
            return out, P
        gamma, beta = ln
        p_drop = cfg["ln_p"]
        seed2 = next_seed() if p_drop > 0 else 0
        sub, y, mean, rstd = L.rowpanel_gemm_ln(O, derived_weight("fm", (Wo,)), bo.detach(), q_in, gamma.detach(),
                                                beta.detach(), cfg["ln_row_pad"], cfg["ln_eps"], p_drop, seed2)
        ctx.save_for_backward(q_in, kv_in, Q, K, V, O, lse, Wq, Wk, Wv, Wo, bq, bk, bv, sub, mean, rstd, gamma)
        ctx.ln_seed = seed2
        return y, P


def _mha_backward(ctx, dout, dres, dbo):
    """MHAFn.backward from the gradient of the output projection's result: dout (contiguous), dres the
    residual-branch gradient the input gradient is accumulated into (or None), dbo the output bias gradient when the
    LayerNorm backward already summed it (else it is taken from dout)."""
    if True:
        q_in, kv_in, Q, K, V, O, lse, Wq, Wk, Wv, Wo, bq, bk, bv = ctx.saved_tensors[:14]
        cfg = ctx.cfg
        B, Lq, Lk, H, dk, dv = cfg["B"], cfg["Lq"], cfg["Lk"], cfg["H"], cfg["dk"], cfg["dv"]
        dt, dev = q_in.dtype, q_in.device
        d = q_in.shape[1]
        Tq, Tk = q_in.shape[0], kv_in.shape[0]
        nq, nk, nv = H * dk, H * dk, H * dv
        ldq, ldk, ldv = ctx.lds
        f32 = torch.float32
        need = ctx.needs_input_grad
        # output projection
        dO = _input_grad(dout, d, (Wo,), Tq, nv, dt)
        dWo, _ = _weight_grad(dout, O, Wo, Tq, need[8])
        if dbo is None:
            dbo = _bias_grad(dout, Tq, d)
        # attention core
        if cfg["self_attn"]:
            dqkv = torch.empty((Tq, nq + nk + nv), dtype=dt, device=dev)
            dQ, dK, dV = dqkv[:, :nq], dqkv[:, nq:nq + nk], dqkv[:, nq + nk:]
            lddq = lddk = lddv = dqkv.stride(0)
        elif cfg.get("kv_pre") is not None:
            # the key / value gradients of all layers share one buffer, which CrossKVFn.backward turns into the
            # encoder-output gradient and the k / v weight gradients once every layer has written its slice
            pre = cfg["kv_pre"]
            dkv_all = pre.shared.grad_buffer(kv_in)
            dQ = torch.empty((Tq, nq), dtype=dt, device=dev)
            dK, dV = dkv_all[:, pre.col:pre.col + nk], dkv_all[:, pre.col + nk:pre.col + nk + nv]
            lddq, lddk, lddv = nq, dkv_all.stride(0), dkv_all.stride(0)
            pre.shared.written.add(pre.col)
        else:
            dQ = torch.empty((Tq, nq), dtype=dt, device=dev)
            dkv = torch.empty((Tk, nk + nv), dtype=dt, device=dev)
            dK, dV = dkv[:, :nk], dkv[:, nk:]
            lddq, lddk, lddv = nq, nk + nv, nk + nv
        delta = torch.empty((B, H, Lq), dtype=f32, device=dev)
        L.attn_bwd(Q, K, V, O, dO, lse, dQ, dK, dV, delta, B, H, Lq, Lk, dk, dv, ldq, ldk, ldv, nv, lddq, lddk, lddv,
                   cfg["scale"], key_pad=cfg["key_pad"], dense_mask=cfg["dense_mask"], causal=cfg["causal"],
                   p_drop=cfg["p_attn"], seed=ctx.seed, q_off=cfg.get("q_off"), k_off=cfg.get("k_off"))
        # projections: weight gradients, bias gradients (column sums) and input gradients.  Queued for a grouped
        # launch (bf16 inside batched_weight_grads()): one problem per parameter, each written straight into its
        # .grad together with its bias gradient.  Otherwise one product and one column sum over the widest buffer
        # that shares an input ([dq|dk|dv] for self-attention, [dk|dv] otherwise).
        if cfg.get("kv_pre") is not None:  # k / v weights and the encoder output get their gradients from CrossKVFn
            dWq, dbq = _weight_grad(dQ, q_in, Wq, Tq, need[2], bias=bq if need[3] else None)
            dq_in = _input_grad(dQ, lddq, (Wq,), Tq, d, dt, add_to=dres)  # accumulates into dres in place
            return dq_in, None, dWq, dbq, None, None, None, None, dWo, dbo, None
        kv_src = q_in if cfg["self_attn"] else kv_in
        batched = dt == torch.bfloat16 and _batchable(dQ, q_in) and _batchable(dK, kv_src) and _batchable(dV, kv_src)
        if batched:
            dWq, dbq = _weight_grad(dQ, q_in, Wq, Tq, need[2], bias=bq if need[3] else None)
            dWk, dbk = _weight_grad(dK, kv_src, Wk, Tk, need[4], bias=bk if need[5] else None)
            dWv, dbv = _weight_grad(dV, kv_src, Wv, Tk, need[6], bias=bv if need[7] else None)
        if cfg["self_attn"]:
            n3 = nq + nk + nv
            if not batched:
                dW = L.gemm(L.GEMM_TN, dqkv, q_in, n3, d, Tq, lddq, q_in.stride(0), out_dtype=f32)
                dWq, dWk, dWv = dW[:nq], dW[nq:nq + nk], dW[nq + nk:]
                db = L.colsum(dqkv, Tq, n3, lddq)
                dbq, dbk, dbv = db[:nq], db[nq:nq + nk], db[nq + nk:]
            dq_in = _input_grad(dqkv, lddq, (Wq, Wk, Wv), Tq, d, dt, add_to=dres)  # accumulates into dres in place
            dkv_in = None
        else:
            if not batched:
                dWq, dbq = _weight_grad(dQ, q_in, Wq, Tq, need[2], bias=bq)
                dW = L.gemm(L.GEMM_TN, dkv, kv_in, nk + nv, d, Tk, lddk, kv_in.stride(0), out_dtype=f32)
                dWk, dWv = dW[:nk], dW[nk:]
                db = L.colsum(dkv, Tk, nk + nv, lddk)
                dbk, dbv = db[:nk], db[nk:]
            dq_in = _input_grad(dQ, lddq, (Wq,), Tq, d, dt, add_to=dres)  # accumulates into dres in place
            dkv_in = _input_grad(dkv, lddk, (Wk, Wv), Tk, d, dt)
        return dq_in, dkv_in, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo, None


class MHALNFn(torch.autograd.Function):
    """MHAFn followed by the sub-layer tail `layer_norm(dropout(fc(output)) + residual)` with pad rows zeroed
    (lib/transformer/SubLayers.py:62-63, Layers.py:22,50,55): the output projection, dropout, residual (= q_in),
    LayerNorm and row mask are ONE launch (skhip_rowpanel_gemm_ln).  cfg carries ln_row_pad, ln_eps, ln_p besides
    MHAFn's entries.  Returns (LayerNorm output (B*Lq, d), attn or None)."""

    @staticmethod
    def forward(ctx, q_in, kv_in, Wq, bq, Wk, bk, Wv, bv, Wo, bo, gamma, beta, cfg):
        return _mha_forward(ctx, q_in, kv_in, Wq, bq, Wk, bk, Wv, bv, Wo, bo, cfg, (gamma, beta))

    @staticmethod
    def backward(ctx, dy, _dP):
        q_in = ctx.saved_tensors[0]
        sub, mean, rstd, gamma = ctx.saved_tensors[14:18]
        cfg = ctx.cfg
        _check_residual_handoff()
        dx, dres, dg, db, dxs = L.add_ln_bwd(dy.contiguous(), sub, q_in, gamma.detach(), cfg["ln_row_pad"], mean, rstd,
                                             cfg["ln_p"], ctx.ln_seed, want_dx_colsum=True)
        grads = _mha_backward(ctx, dx, _unalias(dres, dx), dxs)
        return grads[:10] + (dg, db, None)


def mha_ln_fusable(q_in, Wo, n_v):
    """Whether MHALNFn's one-launch tail applies: bf16, d_model = h * d_v = 512, row count that fills the chip."""
    return (q_in.dtype == torch.bfloat16 and q_in.shape[1] == 512 and n_v == 512 and tuple(Wo.shape) == (512, 512)
            and _panel_rows_ok(q_in, q_in.shape[0]))


class CrossKVShared:
    """What the decoder layers of one forward pass share about their hoisted key / value projections: the buffer
    their backward passes write the key / value gradients into, and which column blocks have been written."""

    def __init__(self):
        self.dkv, self.written = None, set()

    def grad_buffer(self, kv_all):
        if self.dkv is None:
            self.dkv = torch.empty(kv_all.shape, dtype=kv_all.dtype, device=kv_all.device)
        return self.dkv


class CrossKVPre:
    """A decoder layer's handle on the hoisted projections: its column offset in the shared buffers."""

    def __init__(self, shared, col):
        self.shared, self.col = shared, col


class CrossKVFn(torch.autograd.Function):
    """The key | value projections of the encoder output for ALL decoder layers as one product
    (lib/transformer/SubLayers.py:47-48 applied by every DecoderLayer.enc_attn to the same enc_output,
    lib/transformer/Layers.py:50-52): (T_src, d) x [Wk_0; Wv_0; Wk_1; ...]^T -> (T_src, sum n_i), each layer reading
    its column block in place.  Backward runs once, after every layer has left its key / value gradients in the
    shared buffer (they reach it directly, not through autograd, which would add full-width tensors layer by
    layer): one long-K product for the encoder-output gradient instead of one short product and one addition per
    layer, and the weight / bias gradients per parameter (queued for the grouped launch where that is on).
    params: Wk_0, bk_0, Wv_0, bv_0, Wk_1, ..."""

    @staticmethod
    def forward(ctx, enc, shared, *params):
        Ws, bs = params[0::2], params[1::2]
        Tk = enc.shape[0]
        n_tot = sum(_as2d(w)[0] for w in Ws)
        kv = torch.empty((Tk, n_tot), dtype=enc.dtype, device=enc.device)
        _project(enc, Ws, bs, kv, n_tot, enc.dtype)
        ctx.save_for_backward(enc, *params)
        ctx.shared = shared
        ctx.set_materialize_grads(False)
        return kv

    @staticmethod
    def backward(ctx, dkv_direct):
        enc, *params = ctx.saved_tensors
        Ws, bs = params[0::2], params[1::2]
        shared, need = ctx.shared, ctx.needs_input_grad
        Tk, d = enc.shape
        dt = enc.dtype
        n_tot = sum(_as2d(w)[0] for w in Ws)
        dkv = shared.dkv
        shared.dkv = None
        if dkv is None and dkv_direct is None:
            return (None,) * (2 + len(params))
        if dkv is None:
            dkv = dkv_direct.contiguous()
        else:
            # column blocks no layer wrote (a layer outside the differentiated graph) carry no gradient
            off = 0
            for i in range(0, len(Ws), 2):
                n = _as2d(Ws[i])[0] + _as2d(Ws[i + 1])[0]
                if off not in shared.written:
                    dkv[:, off:off + n].zero_()
                off += n
            if dkv_direct is not None:  # the projections were also used through autograd
                dkv = dkv + dkv_direct
        shared.written = set()
        d_enc = _input_grad(dkv, n_tot, Ws, Tk, d, dt) if need[0] else None
        grads, off = [], 0
        for i, (W, b) in enumerate(zip(Ws, bs)):
            n = _as2d(W)[0]
            blk = dkv[:, off:off + n]
            nW, nb = need[2 + 2 * i], need[3 + 2 * i]
            dW, db = _weight_grad(blk, enc, W, Tk, nW, bias=b if nb else None)
            grads += [dW, db]
            off += n
        return (d_enc, None, *grads)


def cross_kv(enc, attns):
    """(kv_all, [CrossKVPre per attention module]) for the encoder-decoder attention modules `attns` of a decoder
    stack, or (None, None) when SK_CROSS_KV_HOIST=0 asks for one projection per layer."""
    if os.environ.get("SK_CROSS_KV_HOIST", "1") == "0" or len(attns) < 2:
        return None, None
    shared, params, pres, col = CrossKVShared(), [], [], 0
    for a in attns:
        params += [a.w_ks.weight, a.w_ks.bias, a.w_vs.weight, a.w_vs.bias]
        pres.append(CrossKVPre(shared, col))
        col += a.w_ks.weight.shape[0] + a.w_vs.weight.shape[0]
    return CrossKVFn.apply(enc, shared, *params), pres


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
    def forward(ctx, x, res, gamma, beta, row_pad, eps, p_drop, res_feeds_x):
        seed = next_seed() if p_drop > 0 else 0
        x, res = x.contiguous(), res.contiguous()
        y, mean, rstd = L.add_ln_fwd(x, res, gamma.detach(), beta.detach(), row_pad, eps, p_drop, seed)
        ctx.save_for_backward(x, res, gamma, mean, rstd)
        ctx.row_pad, ctx.p_drop, ctx.seed = row_pad, p_drop, seed
        ctx.res_feeds_x = res_feeds_x and ctx.needs_input_grad[1]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, gamma, mean, rstd = ctx.saved_tensors
        _check_residual_handoff()
        dx, dres, dg, db, dxs = L.add_ln_bwd(dy.contiguous(), x, res, gamma.detach(), ctx.row_pad, mean, rstd,
                                             ctx.p_drop, ctx.seed, want_dx_colsum=True)
        if dx.data_ptr() == dres.data_ptr():
            dx = dx.view(dx.shape)  # distinct tensor object for the x-branch gradient
        dx._sk_colsum = (dx.data_ptr(), dxs)  # bias gradient of the linear layer that produced x
        if ctx.res_feeds_x:
            # x = f(res) with f an MHAFn / FFNFn: hand the residual-branch gradient to f's backward,
            # whose last GEMM accumulates the input gradient into it, instead of returning it here
            # and letting autograd add the two (T, d) tensors in a separate pass.
            dx._sk_dres = (dx.data_ptr(), dres)
            _pending_dres[dx.data_ptr()] = True
            dres = None
        return dx, dres, dg, db, None, None, None, None


class _InferCtx:
    """Stands in for the autograd context while gradients are off."""
    needs_input_grad = (False,) * 32

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def call(fn, *args):
    """fn.apply(*args); with gradients off, fn.forward on a stand-in context: Function.apply costs some 20 us a call
    there for nothing, a quarter of the host time of an op-by-op inference pass over a few thousand tokens (the encoder
    pass that opens a beam search)."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_InferCtx(), *args)


def add_ln(x, res, gamma, beta, row_pad=None, eps=1e-5, p_drop=0.0, res_feeds_x=False):
    """res_feeds_x=True promises that x was produced from res by ONE MHAFn (as its q_in) or FFNFn call
    and is used by nothing else; the residual gradient is then folded into that call's backward."""
    return call(AddLNFn, x, res, gamma, beta, row_pad, eps, p_drop, res_feeds_x)


# ----------------------------------------------------------------------------- embeddings
class EmbedFn(torch.autograd.Function):
    """word_emb[seq] + pos_table[pos]  (lib/transformer/Models.py:117-118,181-182)."""

    @staticmethod
    def forward(ctx, seq, pos, word_emb, pos_table, dtype):
        seq, pos = seq.contiguous().view(-1), pos.contiguous().view(-1)
        out = L.embed_fwd(seq, pos, word_emb.detach(), pos_table.detach(), dtype)
        ctx.save_for_backward(seq)
        ctx.shape, ctx.table = word_emb.shape, word_emb
        return out

    @staticmethod
    def backward(ctx, dout):
        (seq,) = ctx.saved_tensors
        _check_residual_handoff()
        view, acc = _claim_grad(ctx.table) if ctx.needs_input_grad[2] else (None, False)
        if view is not None:
            if not acc:
                view.zero_()
            L.embed_bwd(seq, dout.contiguous(), view, PAD)
            return None, None, (None if acc else view.view(ctx.shape)), None, None
        dW = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        L.embed_bwd(seq, dout.contiguous(), dW, PAD)
        return None, None, dW, None, None


# ----------------------------------------------------------------------------- loss
class XentFn(torch.autograd.Function):
    """Label-smoothed / plain cross entropy, summed over non-pad rows
    (lib/TransformerModel.py:353-375).  One fused pass yields loss, accuracy counts and the
    gradient; backward only scales the stored gradient."""

    @staticmethod
    def forward(ctx, logits, gold, eps, logit_scale, stats, unit_grad, grad_scale_dev=None):
        logits = logits.contiguous()
        ctx.unit_grad = unit_grad
        want_grad = ctx.needs_input_grad[0]
        dlogits = torch.empty_like(logits) if want_grad else None
        L.xent_fwd_bwd(logits, gold.contiguous().view(-1), dlogits, stats, logit_scale, eps, PAD,
                       grad_scale_dev=grad_scale_dev)
        ctx.dlogits = dlogits
        return stats[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        d = ctx.dlogits
        ctx.dlogits = None
        if not ctx.unit_grad:
            d = d * g.to(d.dtype)
        return d, None, None, None, None, None, None


def xent(logits, gold, smoothing, logit_scale=1.0, unit_grad=False, grad_scale_dev=None):
    """Returns (loss, stats) with stats = float64 [loss_sum, n_correct, n_non_pad] on device.
    unit_grad=True promises that the loss is differentiated directly (loss.backward()), which
    lets backward hand out the stored gradient without a scaling pass over (T,V).
    grad_scale_dev: one-element fp32 device tensor, treated as a constant, that multiplies the gradient
    (not the returned loss) inside the kernel — the classifier loss of train_decoder.py:471."""
    stats = torch.zeros(3, dtype=torch.float64, device=logits.device)
    loss = XentFn.apply(logits, gold, 0.1 if smoothing else 0.0, logit_scale, stats, unit_grad, grad_scale_dev)
    return loss, stats
