"""ctypes binding of libspecialk_hip.so (the C ABI declared in include/specialk_hip.h).

PyTorch is only the host here: it owns device memory and streams.  Every function below
passes raw device pointers, sizes and the current HIP stream to the library.  There is no
CPU or torch fallback: if the shared library is missing, or a call is made with non-GPU
tensors, this module raises.
"""
import ctypes
import os

import torch

SK_F32, SK_BF16 = 0, 1
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_RELU, EPI_RELU_MASK, EPI_ACCUM, EPI_RELU_BITS_OUT, EPI_RELU_MASK_BITS, EPI_ROWSTATS = 1, 2, 4, 8, 16, 32, 64

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKHIP_LIB") or os.path.join(_HERE, "libspecialk_hip.so")  # (SKHIP_LIB: a diagnostics build)

# every symbol include/specialk_hip.h declares
EXPORTS = [
    "skhip_version", "skhip_last_error", "skhip_init", "skhip_gemm", "skhip_gemm_grouped_tn", "skhip_colsum", "skhip_colsum_noalloc", "skhip_attn_fwd",
    "skhip_attn_bwd", "skhip_attn_varlen_fwd", "skhip_attn_varlen_bwd", "skhip_add_ln_fwd", "skhip_add_ln_bwd_ws_floats", "skhip_add_ln_bwd", "skhip_embed_fwd",
    "skhip_embed_bwd", "skhip_pack_plan", "skhip_pack_plans", "skhip_xent_fwd_bwd", "skhip_adam_multi", "skhip_cast_multi", "skhip_cast2d_multi", "skhip_convert",
    "skhip_decode_attn", "skhip_beam_topk_ws_bytes", "skhip_beam_topk", "skhip_beam_update", "skhip_gather_rows",
    "skhip_decode_stack_ws_bytes", "skhip_decode_stack", "skhip_beam_topk_stats", "skhip_beam_step",
    "skhip_softmax_rows", "skhip_conv_windows", "skhip_cnn_head", "skhip_collate", "skhip_set_reserved_cus",
    "skhip_rowpanel_rows", "skhip_rowpanel_pattern_bytes", "skhip_rowpanel_set_stamps", "skhip_rowpanel_gemm", "skhip_rowpanel_gemm_ln",
    "skhip_generator_chunk_cols", "skhip_generator_rowstats", "skhip_beam_backtrack",
]

_lib = None
c_i64, c_int, c_f32, c_u64, c_p = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, ctypes.c_void_p


def load():
    """Load the library once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C specialk_amd/csrc` (or __graft_entry__.build()). "
            "specialk_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.skhip_version.restype = ctypes.c_char_p
    lib.skhip_last_error.restype = ctypes.c_char_p
    lib.skhip_add_ln_bwd_ws_floats.restype = c_i64
    lib.skhip_add_ln_bwd_ws_floats.argtypes = [c_int]
    lib.skhip_beam_topk_ws_bytes.restype = c_i64
    lib.skhip_beam_topk_ws_bytes.argtypes = [c_int, c_int, c_i64]
    lib.skhip_pack_plan.argtypes = [c_p, c_int, c_int, c_int, c_i64, c_p, c_p, c_p, c_p, c_p]
    lib.skhip_pack_plans.argtypes = [c_p, c_int, c_i64, c_p, c_p]
    lib.skhip_collate.argtypes = [c_p, c_p, c_int, c_int, c_i64, c_p, c_p, c_p]
    lib.skhip_gemm_grouped_tn.argtypes = [ctypes.POINTER(c_i64), c_int, c_p, c_i64, c_p]
    lib.skhip_gemm.argtypes = [c_int, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32,
                               c_int, c_int, c_int, c_p, c_i64, c_p]
    lib.skhip_colsum.argtypes = [c_p, c_p, c_i64, c_i64, c_i64, c_int, c_int, c_p, c_i64, c_p]
    lib.skhip_colsum_noalloc.argtypes = [c_p, c_p, c_i64, c_i64, c_i64, c_int, c_int, c_p]
    lib.skhip_attn_fwd.argtypes = [c_p] * 8 + [c_int] * 6 + [c_i64] * 4 + [c_f32, c_int, c_f32, c_u64, c_int, c_p]
    lib.skhip_attn_bwd.argtypes = [c_p] * 12 + [c_int] * 6 + [c_i64] * 7 + [c_f32, c_int, c_f32, c_u64, c_int, c_p]
    lib.skhip_attn_varlen_fwd.argtypes = [c_p] * 7 + [c_int] * 6 + [c_i64] * 4 + [c_f32, c_int, c_f32, c_u64, c_int, c_p]
    lib.skhip_attn_varlen_bwd.argtypes = [c_p] * 12 + [c_int] * 6 + [c_i64] * 7 + [c_f32, c_int, c_f32, c_u64, c_int, c_p]
    lib.skhip_add_ln_fwd.argtypes = [c_p] * 8 + [c_i64, c_int, c_f32, c_f32, c_u64, c_int, c_p]
    lib.skhip_add_ln_bwd.argtypes = [c_p] * 13 + [c_i64, c_int, c_f32, c_u64, c_int, c_int, c_p]
    lib.skhip_embed_fwd.argtypes = [c_p] * 5 + [c_i64, c_int, c_i64, c_i64, c_int, c_p]
    lib.skhip_embed_bwd.argtypes = [c_p] * 3 + [c_i64, c_int, c_i64, c_i64, c_int, c_p]
    lib.skhip_xent_fwd_bwd.argtypes = [c_p] * 4 + [c_i64, c_i64, c_i64, c_f32, c_f32, c_i64, c_f32, c_p, c_int, c_p]
    lib.skhip_softmax_rows.argtypes = [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_p]
    lib.skhip_conv_windows.argtypes = [c_p, c_p, c_p] + [c_int] * 5 + [c_i64] * 6 + [c_int, c_p]
    lib.skhip_cnn_head.argtypes = [c_p] * 6 + [c_int, c_int, c_int, c_i64, c_int, c_int, c_p]
    lib.skhip_adam_multi.argtypes = [c_p] * 6 + [c_int, c_i64, c_f32, c_f32, c_f32, c_f32, c_int, c_f32, c_p]
    lib.skhip_cast_multi.argtypes = [c_p] * 3 + [c_int, c_i64, c_p]
    lib.skhip_cast2d_multi.argtypes = [c_p] * 3 + [c_int, c_i64, c_p]
    lib.skhip_convert.argtypes = [c_p, c_p, c_i64, c_int, c_int, c_p]
    lib.skhip_decode_attn.argtypes = [c_p] * 9 + [c_i64] + [c_int] * 6 + [c_i64, c_i64, c_f32, c_int, c_int, c_p]
    lib.skhip_beam_topk.argtypes = [c_p] * 6 + [c_int, c_int, c_i64, c_int, c_int, c_p]
    lib.skhip_beam_topk_stats.argtypes = [c_p] * 7 + [c_int, c_int, c_i64, c_int, c_int, c_p]
    lib.skhip_beam_update.argtypes = [c_p] * 16 + [c_int] * 6 + [c_p]
    lib.skhip_beam_step.argtypes = [c_p] * 19 + [c_int, c_int, c_i64] + [c_int] * 6 + [c_p]
    lib.skhip_decode_stack_ws_bytes.argtypes = [c_int]
    lib.skhip_decode_stack_ws_bytes.restype = c_i64
    lib.skhip_decode_stack.argtypes = [c_p, c_int] + [c_p] * 11 + [c_i64] + [c_int] * 8 + [c_f32, c_f32, c_int, c_int, c_p]
    lib.skhip_gather_rows.argtypes = [c_p] * 3 + [c_i64] * 4 + [c_int, c_p]
    lib.skhip_rowpanel_set_stamps.argtypes = [c_p]
    lib.skhip_rowpanel_pattern_bytes.argtypes = [c_i64, c_i64]
    lib.skhip_rowpanel_pattern_bytes.restype = c_i64
    lib.skhip_rowpanel_gemm.argtypes = [c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_int, c_p]
    lib.skhip_rowpanel_gemm_ln.argtypes = [c_p, c_i64, c_p, c_p, c_p, c_i64] + [c_p] * 7 + [c_i64, c_f32, c_f32, c_u64, c_p]
    lib.skhip_generator_rowstats.argtypes = [c_p, c_i64, c_int, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p]
    lib.skhip_beam_backtrack.argtypes = [c_p] * 4 + [c_int] * 4 + [c_i64, c_p]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError here means header and library disagree
    _lib = lib
    return lib


def version():
    return load().skhip_version().decode()


reserved_cus = int(os.environ.get("SKHIP_RESERVE_CUS", "0") or 0)   # what the library currently plans without


def set_reserved_cus(n):
    """CUs the large-tile GEMMs leave out of their round planning (held by communication kernels meanwhile)."""
    global reserved_cus
    _check(load().skhip_set_reserved_cus(int(n)), "skhip_set_reserved_cus")
    reserved_cus = int(n)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().skhip_last_error().decode()}")


def dcode(dtype):
    if dtype == torch.float32:
        return SK_F32
    if dtype == torch.bfloat16:
        return SK_BF16
    raise TypeError(f"specialk_amd supports float32 and bfloat16 storage, got {dtype}")


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("specialk_amd kernels need GPU tensors (there is no CPU path)")
    return t.data_ptr()


# The current stream's handle, per launch: torch.cuda.current_stream() builds a Stream object and resolves the device
# three times over (7 us a call, a quarter of the host time of a short op-by-op pass); the raw accessors behind it take
# well under one.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_workspaces = {}


def workspace(nbytes=64 << 20, device=None):
    """Persistent scratch per device and stream (split-K slabs, LayerNorm partials, beam candidates):
    kernels enqueued on different streams may run concurrently and must not share scratch."""
    if device is None and _raw_stream is not None and _raw_device is not None:
        index = _raw_device()
        key = (index, _raw_stream(index))
        ws = _workspaces.get(key)
        if ws is not None and ws.numel() >= nbytes:
            return ws
        device = torch.device("cuda", index)
    else:
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        index = device.index if device.index is not None else torch.cuda.current_device()
        key = (index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# --------------------------------------------------------------------------- GEMM
def gemm(layout, A, B, M, N, K, lda, ldb, out=None, ldc=None, bias=None, aux=None, alpha=1.0, epilogue=0,
         out_dtype=None, split_k=True):
    """C[M,N] = alpha * op(A) op(B) with the library's epilogues.  A, B, out are 2-D views whose
    base pointer and leading dimension are given explicitly (so slices of fused buffers work)."""
    lib = load()
    out_dtype = A.dtype if out_dtype is None else out_dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
        ldc = N
    if bias is not None:
        assert bias.dtype == torch.float32
        epilogue |= EPI_BIAS
    ws = workspace() if split_k else None
    rc = lib.skhip_gemm(layout, _ptr(A), _ptr(B), _ptr(out), _ptr(bias), _ptr(aux), M, N, K, lda, ldb, ldc,
                        float(alpha), epilogue, dcode(A.dtype), dcode(out.dtype), _ptr(ws),
                        ws.numel() if ws is not None else 0, _stream())
    _check(rc, "skhip_gemm")
    return out


def gemm_grouped_tn(problems):
    """problems: list of (A, B, C, M, N, K, lda, ldb, ldc, accumulate[, colsum, colsum_accumulate]) with A (K, M) and
    B (K, N) bf16 views (rows contiguous), C (M, N) fp32: C (+)= A^T B for all of them in one pass over the chip;
    colsum (optional fp32 (M,)) (+)= column sums of A."""
    lib = load()
    n = len(problems)
    if n == 0:
        return
    table = (c_i64 * (12 * n))()
    for i, pr in enumerate(problems):
        A, B, C, M, N, K, lda, ldb, ldc, acc = pr[:10]
        cs, cs_acc = (pr[10], pr[11]) if len(pr) > 10 else (None, False)
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and C.dtype == torch.float32
        assert cs is None or (cs.dtype == torch.float32 and cs.is_contiguous() and cs.numel() == M)
        table[12 * i:12 * i + 12] = [A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, lda, ldb, ldc, int(bool(acc)),
                                     cs.data_ptr() if cs is not None else 0, int(bool(cs_acc))]
    ws = workspace(192 << 20)  # up to two rounds of 256 CUs x 256 KB tile slabs, with room to spare
    _check(lib.skhip_gemm_grouped_tn(table, n, _ptr(ws), ws.numel(), _stream()), "skhip_gemm_grouped_tn")


def colsum(X, M, N, ldx, out=None, accumulate=False):
    lib = load()
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    ws = workspace()
    _check(lib.skhip_colsum(_ptr(X), _ptr(out), M, N, ldx, int(accumulate), dcode(X.dtype), _ptr(ws), ws.numel(),
                            _stream()), "skhip_colsum")
    return out


def colsum_noalloc(X, M, N, ldx, out, accumulate=False):
    """skhip_colsum with the partial sums in the library's built-in per-GPU scratch array: the call allocates
    nothing (out is the caller's).  Calls that overlap on one GPU race on that array."""
    _check(load().skhip_colsum_noalloc(_ptr(X), _ptr(out), M, N, ldx, int(accumulate), dcode(X.dtype), _stream()),
           "skhip_colsum_noalloc")
    return out


# --------------------------------------------------------------------------- row-panel products (d_model = 512, bf16)
ROWPANEL_ROWS = 104


def rowpanel_set_stamps(buf):
    """Diagnostics: int64 device tensor of ceil(T / 104) * 8 * 32 entries that the row-panel kernels fill with clock
    readings (None: off)."""
    _check(load().skhip_rowpanel_set_stamps(_ptr(buf)), "skhip_rowpanel_set_stamps")


def rowpanel_pattern(T, N, device):
    """Buffer for the ReLU pattern of a (T, N) row-panel product (the kernels' own layout)."""
    return torch.empty(int(load().skhip_rowpanel_pattern_bytes(T, N)) // 4, dtype=torch.int32, device=device)


def rowpanel_gemm(a, w_fm, N, bias=None, out=None, ldc=None, signs=None, epilogue=0):
    """out (T, N) = epilogue(a (T, 512; any row stride) w^T) with w (N, 512) fragment-major (ops.derived_weight "fm")."""
    T = a.shape[0]
    assert a.dtype == torch.bfloat16 and a.shape[1] == 512 and a.stride(1) == 1
    if out is None:
        out, ldc = torch.empty((T, N), dtype=torch.bfloat16, device=a.device), N
    if bias is not None:
        epilogue |= EPI_BIAS
    _check(load().skhip_rowpanel_gemm(_ptr(a), a.stride(0), _ptr(w_fm), _ptr(bias), _ptr(out), ldc, _ptr(signs), T, N,
                                      epilogue, _stream()), "skhip_rowpanel_gemm")
    return out


def rowpanel_gemm_ln(a, w_fm, bias, res, gamma, beta, row_pad, eps, p_drop, seed):
    """(sub_out, y, mean, rstd): sub_out = a w^T + bias (T, 512), y = LayerNorm(res + dropout(sub_out)) with pad rows
    zeroed — add_ln_fwd fused behind the product (w (512, 512) fragment-major)."""
    T = a.shape[0]
    dev = a.device
    assert a.dtype == torch.bfloat16 and a.shape[1] == 512 and a.stride(1) == 1 and res.stride(1) == 1
    sub = torch.empty((T, 512), dtype=torch.bfloat16, device=dev)
    y = torch.empty((T, 512), dtype=torch.bfloat16, device=dev)
    mean = torch.empty(T, dtype=torch.float32, device=dev)
    rstd = torch.empty(T, dtype=torch.float32, device=dev)
    rc = load().skhip_rowpanel_gemm_ln(_ptr(a), a.stride(0), _ptr(w_fm), _ptr(bias), _ptr(res), res.stride(0), _ptr(sub),
                                       _ptr(gamma), _ptr(beta), _ptr(row_pad), _ptr(y), _ptr(mean), _ptr(rstd), T,
                                       float(eps), float(p_drop), int(seed), _stream())
    _check(rc, "skhip_rowpanel_gemm_ln")
    return sub, y, mean, rstd


# --------------------------------------------------------------------------- attention
def attn_fwd(Q, K, V, O, LSE, B, H, Lq, Lk, dk, dv, ldq, ldk, ldv, ldo, scale, key_pad=None, dense_mask=None,
             causal=False, p_drop=0.0, seed=0, P_out=None, q_off=None, k_off=None):
    """q_off / k_off (int32 [B+1] row offsets) select the packed form; Lq, Lk are then the maxima."""
    lib = load()
    if q_off is not None:
        assert key_pad is None and dense_mask is None and P_out is None and q_off.dtype == torch.int32
        rc = lib.skhip_attn_varlen_fwd(_ptr(Q), _ptr(K), _ptr(V), _ptr(O), _ptr(LSE), _ptr(q_off), _ptr(k_off), B, H,
                                       Lq, Lk, dk, dv, ldq, ldk, ldv, ldo, float(scale), int(causal), float(p_drop),
                                       int(seed), dcode(Q.dtype), _stream())
        return _check(rc, "skhip_attn_varlen_fwd")
    rc = lib.skhip_attn_fwd(_ptr(Q), _ptr(K), _ptr(V), _ptr(O), _ptr(LSE), _ptr(P_out), _ptr(key_pad),
                            _ptr(dense_mask), B, H, Lq, Lk, dk, dv, ldq, ldk, ldv, ldo, float(scale), int(causal),
                            float(p_drop), int(seed), dcode(Q.dtype), _stream())
    _check(rc, "skhip_attn_fwd")


def attn_bwd(Q, K, V, O, dO, LSE, dQ, dK, dV, delta, B, H, Lq, Lk, dk, dv, ldq, ldk, ldv, ldo, lddq, lddk, lddv,
             scale, key_pad=None, dense_mask=None, causal=False, p_drop=0.0, seed=0, q_off=None, k_off=None):
    lib = load()
    if q_off is not None:
        assert key_pad is None and dense_mask is None and q_off.dtype == torch.int32
        rc = lib.skhip_attn_varlen_bwd(_ptr(Q), _ptr(K), _ptr(V), _ptr(O), _ptr(dO), _ptr(LSE), _ptr(dQ), _ptr(dK),
                                       _ptr(dV), _ptr(delta), _ptr(q_off), _ptr(k_off), B, H, Lq, Lk, dk, dv, ldq, ldk,
                                       ldv, ldo, lddq, lddk, lddv, float(scale), int(causal), float(p_drop), int(seed),
                                       dcode(Q.dtype), _stream())
        return _check(rc, "skhip_attn_varlen_bwd")
    rc = lib.skhip_attn_bwd(_ptr(Q), _ptr(K), _ptr(V), _ptr(O), _ptr(dO), _ptr(LSE), _ptr(dQ), _ptr(dK), _ptr(dV),
                            _ptr(delta), _ptr(key_pad), _ptr(dense_mask), B, H, Lq, Lk, dk, dv, ldq, ldk, ldv, ldo,
                            lddq, lddk, lddv, float(scale), int(causal), float(p_drop), int(seed), dcode(Q.dtype),
                            _stream())
    _check(rc, "skhip_attn_bwd")


# --------------------------------------------------------------------------- add + LayerNorm
def add_ln_fwd(x, res, gamma, beta, row_pad, eps, p_drop, seed):
    lib = load()
    T, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(T, dtype=torch.float32, device=x.device)
    rstd = torch.empty(T, dtype=torch.float32, device=x.device)
    rc = lib.skhip_add_ln_fwd(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(row_pad), _ptr(y), _ptr(mean),
                              _ptr(rstd), T, d, float(eps), float(p_drop), int(seed), dcode(x.dtype), _stream())
    _check(rc, "skhip_add_ln_fwd")
    return y, mean, rstd


def add_ln_bwd(dy, x, res, gamma, row_pad, mean, rstd, p_drop, seed, dgamma=None, dbeta=None, accumulate=False,
               want_dx_colsum=False):
    lib = load()
    T, d = x.shape
    dres = torch.empty_like(x)
    dx = torch.empty_like(x) if p_drop > 0 else dres
    if dgamma is None:
        dgamma = torch.empty(d, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(d, dtype=torch.float32, device=x.device)
        accumulate = False
    dxs = torch.empty(d, dtype=torch.float32, device=x.device) if want_dx_colsum else None
    ws = workspace(int(lib.skhip_add_ln_bwd_ws_floats(d)) * 4)
    rc = lib.skhip_add_ln_bwd(_ptr(dy), _ptr(x), _ptr(res), _ptr(gamma), _ptr(row_pad), _ptr(mean), _ptr(rstd),
                              _ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(dxs), _ptr(ws), T, d,
                              float(p_drop), int(seed), int(accumulate), dcode(x.dtype), _stream())
    _check(rc, "skhip_add_ln_bwd")
    if want_dx_colsum:
        return dx, dres, dgamma, dbeta, dxs
    return dx, dres, dgamma, dbeta


# --------------------------------------------------------------------------- embeddings
def embed_fwd(seq, pos, word_emb, pos_table, dtype):
    lib = load()
    T = seq.numel()
    d = word_emb.shape[1]
    out = torch.empty((T, d), dtype=dtype, device=seq.device)
    rc = lib.skhip_embed_fwd(_ptr(seq), _ptr(pos), _ptr(word_emb), _ptr(pos_table), _ptr(out), T, d,
                             word_emb.shape[0], pos_table.shape[0], dcode(dtype), _stream())
    _check(rc, "skhip_embed_fwd")
    return out


def pack_plan(seq, trim, pad_id):
    """seq (B, L) int64 on the GPU -> (lens int32 (B,), off int32 (B+1,), idx int64 (B*L,), facts int64 (4,))."""
    lib = load()
    B, L = seq.shape
    dev = seq.device
    seq = seq.contiguous()
    lens = torch.empty(B, dtype=torch.int32, device=dev)
    off = torch.empty(B + 1, dtype=torch.int32, device=dev)
    idx = torch.empty(B * L, dtype=torch.int64, device=dev)
    facts = torch.empty(4, dtype=torch.int64, device=dev)
    _check(lib.skhip_pack_plan(_ptr(seq), B, L, int(trim), int(pad_id), _ptr(lens), _ptr(off), _ptr(idx), _ptr(facts),
                               _stream()), "skhip_pack_plan")
    return lens, off, idx, facts


PACK_FACTS = 6


def pack_plans(items, pad_id):
    """Row bookkeeping of several right-padded id batches in one pair of launches (skhip_pack_plans).
    items: [(seq (B, L) int64, pos (B, L) int64 or None, trim, n_vocab (0: ids not checked), n_pos)], rows may be
    strided.  -> ([(lens, off, idx)], facts int64 (len(items), 6) on the device)."""
    lib = load()
    dev = items[0][0].device
    empty = [it[0].numel() == 0 for it in items]
    if any(empty):  # batches without rows or columns: nothing to launch for them
        full = [it for it, e in zip(items, empty) if not e]
        sub_outs, sub_facts = pack_plans(full, pad_id) if full else ([], None)
        facts = torch.tensor([[1, 0x7fffffff, 0, 0, 1, 1]] * len(items), dtype=torch.int64, device=dev)
        outs, j = [], 0
        for k, (it, e) in enumerate(zip(items, empty)):
            if e:
                B = it[0].shape[0]
                outs.append((torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B + 1, dtype=torch.int32, device=dev),
                             torch.zeros(0, dtype=torch.int64, device=dev)))
            else:
                outs.append(sub_outs[j])
                facts[k] = sub_facts[j]
                j += 1
        return outs, facts
    table = (ctypes.c_int64 * (12 * len(items)))()
    outs, keep = [], []
    for k, (seq, pos, trim, n_vocab, n_pos) in enumerate(items):
        B, Ln = seq.shape
        if seq.stride(1) != 1:
            seq = seq.contiguous()
        if pos is not None and pos.stride(1) != 1:
            pos = pos.contiguous()
        assert seq.dtype == torch.int64 and (pos is None or (pos.dtype == torch.int64 and pos.shape == seq.shape))
        lens = torch.empty(B, dtype=torch.int32, device=dev)
        off = torch.empty(B + 1, dtype=torch.int32, device=dev)
        idx = torch.empty(B * Ln, dtype=torch.int64, device=dev)
        keep.append((seq, pos))
        outs.append((lens, off, idx))
        table[12 * k:12 * k + 12] = [seq.data_ptr(), 0 if pos is None else pos.data_ptr(), seq.stride(0),
                                     0 if pos is None else pos.stride(0), B, Ln, int(trim), int(n_vocab), int(n_pos),
                                     lens.data_ptr(), off.data_ptr(), idx.data_ptr()]
    facts = torch.empty((len(items), PACK_FACTS), dtype=torch.int64, device=dev)
    _check(lib.skhip_pack_plans(table, len(items), int(pad_id), _ptr(facts), _stream()), "skhip_pack_plans")
    return outs, facts


def collate(flat, off, B, L, pad_id):
    """flat int64 ids and off int64 (B+1,) on the GPU -> (seq, pos) int64 (B, L), right-padded."""
    lib = load()
    seq = torch.empty((B, L), dtype=torch.int64, device=flat.device)
    pos = torch.empty((B, L), dtype=torch.int64, device=flat.device)
    _check(lib.skhip_collate(_ptr(flat), _ptr(off), B, L, int(pad_id), _ptr(seq), _ptr(pos), _stream()), "skhip_collate")
    return seq, pos


def embed_bwd(seq, dout, dword_emb, pad_id):
    lib = load()
    T = seq.numel()
    d = dword_emb.shape[1]
    rc = lib.skhip_embed_bwd(_ptr(seq), _ptr(dout), _ptr(dword_emb), T, d, dword_emb.shape[0], pad_id,
                             dcode(dout.dtype), _stream())
    _check(rc, "skhip_embed_bwd")


# --------------------------------------------------------------------------- loss
def xent_fwd_bwd(logits, gold, dlogits, stats, logit_scale, smoothing_eps, pad_id, grad_scale=1.0, grad_scale_dev=None):
    """logits (T,V) contiguous; stats float64[3] on device, accumulated into; grad_scale_dev: optional
    one-element fp32 device tensor that multiplies the gradient."""
    lib = load()
    T, V = logits.shape
    if grad_scale_dev is not None:
        assert grad_scale_dev.dtype == torch.float32 and grad_scale_dev.numel() == 1
    rc = lib.skhip_xent_fwd_bwd(_ptr(logits), _ptr(gold), _ptr(dlogits), _ptr(stats), T, V, logits.stride(0),
                                float(logit_scale), float(smoothing_eps), pad_id, float(grad_scale),
                                _ptr(grad_scale_dev), dcode(logits.dtype), _stream())
    _check(rc, "skhip_xent_fwd_bwd")


# --------------------------------------------------------------------------- style-transfer classifier path
def softmax_rows(x, out=None):
    """Row softmax of a 2-D tensor (rows contiguous); out may be x itself."""
    lib = load()
    rows, V = x.shape
    assert x.stride(1) == 1
    if out is None:
        out = torch.empty((rows, V), dtype=x.dtype, device=x.device)
    _check(lib.skhip_softmax_rows(_ptr(x), _ptr(out), rows, V, x.stride(0), out.stride(0), dcode(x.dtype), _stream()),
           "skhip_softmax_rows")
    return out


def conv_windows(emb, B, L, S, filter_size, E, off=None, pad_row=0, ldcols=None, stride_b=None, stride_t=1):
    """emb (rows, >=E) -> cols (B * (S - filter_size + 1), filter_size * E) (see the header)."""
    lib = load()
    P = S - filter_size + 1
    ldcols = filter_size * E if ldcols is None else ldcols
    cols = torch.empty((max(B * P, 0), ldcols), dtype=emb.dtype, device=emb.device)
    if ldcols > filter_size * E:
        cols.zero_()
    _check(lib.skhip_conv_windows(_ptr(emb), _ptr(off), _ptr(cols), B, L, S, filter_size, E, emb.stride(0), ldcols,
                                  emb.shape[0], int(pad_row), L if stride_b is None else stride_b, stride_t,
                                  dcode(emb.dtype), _stream()), "skhip_conv_windows")
    return cols


def cnn_head(h, w, bias, B, P, F, target_label):
    """h (B*P, >=F) ReLU'd convolution outputs -> (y (B,) fp32, loss (1,) fp32)."""
    lib = load()
    y = torch.empty(B, dtype=torch.float32, device=h.device)
    terms = torch.empty(B, dtype=torch.float32, device=h.device)
    loss = torch.empty(1, dtype=torch.float32, device=h.device)
    assert w.dtype == torch.float32 and bias.dtype == torch.float32 and w.is_contiguous()
    _check(lib.skhip_cnn_head(_ptr(h), _ptr(w), _ptr(bias), _ptr(y), _ptr(terms), _ptr(loss), B, P, F, h.stride(0),
                              int(target_label), dcode(h.dtype), _stream()), "skhip_cnn_head")
    return y, loss


# --------------------------------------------------------------------------- optimiser
def adam_multi(tables, n, max_count, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """tables: dict of device int64 tensors 'p','g','m','v','b'(or None),'n'."""
    lib = load()
    rc = lib.skhip_adam_multi(_ptr(tables["p"]), _ptr(tables["g"]), _ptr(tables["m"]), _ptr(tables["v"]),
                              _ptr(tables.get("b")), _ptr(tables["n"]), n, max_count, float(lr), float(beta1),
                              float(beta2), float(eps), int(step), float(grad_scale), _stream())
    _check(rc, "skhip_adam_multi")


def cast_multi(src_table, dst_table, counts, n, max_count):
    lib = load()
    _check(lib.skhip_cast_multi(_ptr(src_table), _ptr(dst_table), _ptr(counts), n, max_count, _stream()),
           "skhip_cast_multi")


def cast2d_multi(src_table, dst_table, meta, n, max_tiles):
    """Tables are int64 device tensors: source / destination pointers (n,), meta (n, 4) = R, C, ld, flags."""
    _check(load().skhip_cast2d_multi(_ptr(src_table), _ptr(dst_table), _ptr(meta), n, max_tiles, _stream()),
           "skhip_cast2d_multi")


def convert(src, dst):
    lib = load()
    assert src.numel() == dst.numel() and src.is_contiguous() and dst.is_contiguous()
    _check(lib.skhip_convert(_ptr(src), _ptr(dst), src.numel(), dcode(src.dtype), dcode(dst.dtype), _stream()),
           "skhip_convert")
    return dst


# --------------------------------------------------------------------------- decode
def decode_attn(q, Kc, Vc, out, R, H, length, Lmax, dk, dv, ldq, ldo, scale, kv_len=None, key_pad=None,
                kv_row_div=1, hist_row=None, new_k=None, new_v=None, ld_new=0):
    """new_k / new_v: rows of the newest position's key / value; the kernel appends them to the caches."""
    lib = load()
    assert hist_row is None or (hist_row.dtype == torch.int32 and hist_row.is_contiguous())
    rc = lib.skhip_decode_attn(_ptr(q), _ptr(Kc), _ptr(Vc), _ptr(out), _ptr(kv_len), _ptr(key_pad), _ptr(hist_row),
                               _ptr(new_k), _ptr(new_v), int(ld_new), R, H, length,
                               Lmax, dk, dv, ldq, ldo, float(scale), int(kv_row_div), dcode(q.dtype), _stream())
    _check(rc, "skhip_decode_attn")


def beam_topk(logits, scores_in, n_sent, bm, first_step, normalize=True, out=None):
    """out: optional (scores_out, prev, tok) to write into."""
    lib = load()
    V = logits.shape[-1]
    dev = logits.device
    if out is not None:
        scores_out, prev, tok = out
    else:
        scores_out = torch.empty((n_sent, bm), dtype=torch.float32, device=dev)
        prev = torch.empty((n_sent, bm), dtype=torch.int32, device=dev)
        tok = torch.empty((n_sent, bm), dtype=torch.int32, device=dev)
    ws = workspace(int(lib.skhip_beam_topk_ws_bytes(n_sent, bm, V)))
    rc = lib.skhip_beam_topk(_ptr(logits), _ptr(scores_in), _ptr(scores_out), _ptr(prev), _ptr(tok), _ptr(ws),
                             n_sent, bm, V, int(first_step), int(normalize), _stream())
    _check(rc, "skhip_beam_topk")
    return scores_out, prev, tok


def generator_chunk_cols():
    return int(load().skhip_generator_chunk_cols())


ROWS_PLAIN, ROWS_FRAGMENT_MAJOR = 0, 1


def fragment_major_rows(x):
    """x [R][K] (K a multiple of 32) in the order skhip_generator_rowstats reads MFMA operands in
    (ROWS_FRAGMENT_MAJOR): a [ceil(R / 16) * 16][K] tensor, rows past R zero."""
    R, K = x.shape
    nb = (R + 15) // 16
    xp = torch.zeros((nb * 16, K), dtype=x.dtype, device=x.device)
    xp[:R] = x
    return xp.view(nb, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(nb * 16, K)


def plain_rows(xfm, R):
    """The inverse of fragment_major_rows: rows [R][K]."""
    n16, K = xfm.shape
    return xfm.view(n16 // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(n16, K)[:R]


def generator_rowstats(x, w, logits, stats, x_layout=ROWS_PLAIN):
    """logits [R][V] fp32 = x [R][512] @ w [V][512]^T (bf16) and the chunk statistics the beam top-k reads
    (stats [R][ceil(V / generator_chunk_cols())][2]).  x_layout ROWS_FRAGMENT_MAJOR: x is fragment_major_rows of the
    rows (what decode_stack writes on request)."""
    R, V = logits.shape
    K = w.shape[1]
    assert x.dtype == w.dtype == torch.bfloat16 and w.is_contiguous() and w.shape[0] == V and x.shape[1] == K
    assert x.stride(1) == 1 and (x_layout == ROWS_PLAIN and x.shape[0] == R or
                                 x_layout == ROWS_FRAGMENT_MAJOR and x.is_contiguous() and x.shape[0] == (R + 15) // 16 * 16)
    assert logits.dtype == stats.dtype == torch.float32 and logits.is_contiguous()
    assert stats.numel() >= R * ((V + generator_chunk_cols() - 1) // generator_chunk_cols()) * 2
    rc = load().skhip_generator_rowstats(_ptr(x), x.stride(0), int(x_layout), _ptr(w), _ptr(logits), _ptr(stats), R, V, K,
                                         _stream())
    _check(rc, "skhip_generator_rowstats")


def beam_topk_stats(logits, stats, scores_in, n_sent, bm, first_step, out, chunk_cols=64):
    """beam_topk for logits whose generator product left chunk statistics (EPI_ROWSTATS: 64 columns per chunk;
    generator_rowstats: generator_chunk_cols()) in `stats`."""
    lib = load()
    V = logits.shape[-1]
    scores_out, prev, tok = out
    assert logits.dtype == stats.dtype == torch.float32
    assert stats.numel() >= n_sent * bm * ((V + chunk_cols - 1) // chunk_cols) * 2
    ws = workspace(int(lib.skhip_beam_topk_ws_bytes(n_sent, bm, V)))
    rc = lib.skhip_beam_topk_stats(_ptr(logits), _ptr(stats), _ptr(scores_in), _ptr(scores_out), _ptr(prev), _ptr(tok),
                                   _ptr(ws), n_sent, bm, V, int(first_step), int(chunk_cols), _stream())
    _check(rc, "skhip_beam_topk_stats")
    return scores_out, prev, tok


def beam_update(new_scores, prev, new_tok, scores, ys, ks, done, length, n_done, cur_len, tok, pos, row_pad, pad_hist,
                hist_in, hist_out, n_sent, bm, Tc, max_len, pad_id, eos_id):
    """The state update between two beam-search steps (see the header); every argument is a device tensor."""
    assert done.dtype == length.dtype == n_done.dtype == cur_len.dtype == torch.int32
    assert tok.dtype == pos.dtype == torch.int64 and row_pad.dtype == pad_hist.dtype == torch.uint8
    rc = load().skhip_beam_update(_ptr(new_scores), _ptr(prev), _ptr(new_tok), _ptr(scores), _ptr(ys), _ptr(ks), _ptr(done),
                                  _ptr(length), _ptr(n_done), _ptr(cur_len), _ptr(tok), _ptr(pos), _ptr(row_pad),
                                  _ptr(pad_hist), _ptr(hist_in), _ptr(hist_out), n_sent, bm, Tc, max_len, pad_id, eos_id,
                                  _stream())
    _check(rc, "skhip_beam_update")


def beam_step(logits, stats, out, scores, ys, ks, done, length, n_done, cur_len, tok, pos, row_pad, pad_hist, hist_in,
              hist_out, n_sent, bm, first_step, Tc, max_len, pad_id, eos_id, chunk_cols=64):
    """beam_topk_stats + beam_update in one launch; out = (new_scores, prev, new_tok) as for beam_topk_stats."""
    lib = load()
    V = logits.shape[-1]
    new_scores, prev, new_tok = out
    assert logits.dtype == stats.dtype == torch.float32
    assert stats.numel() >= n_sent * bm * ((V + chunk_cols - 1) // chunk_cols) * 2
    assert done.dtype == length.dtype == n_done.dtype == cur_len.dtype == torch.int32
    assert tok.dtype == pos.dtype == torch.int64 and row_pad.dtype == pad_hist.dtype == torch.uint8
    ws = workspace(int(lib.skhip_beam_topk_ws_bytes(n_sent, bm, V)))
    rc = lib.skhip_beam_step(_ptr(logits), _ptr(stats), _ptr(new_scores), _ptr(prev), _ptr(new_tok), _ptr(ws), _ptr(scores),
                             _ptr(ys), _ptr(ks), _ptr(done), _ptr(length), _ptr(n_done), _ptr(cur_len), _ptr(tok), _ptr(pos),
                             _ptr(row_pad), _ptr(pad_hist), _ptr(hist_in), _ptr(hist_out), n_sent, bm, V, int(first_step),
                             Tc, max_len, pad_id, eos_id, int(chunk_cols), _stream())
    _check(rc, "skhip_beam_step")


def beam_backtrack(ys, ks, length, n_sent, bm, n_best, steps):
    """Tokens of the n_best hypotheses of every sentence, int32 [n_sent][n_best][steps] (zeros past a hypothesis'
    length): the back-pointer walk of Beam.get_hypothesis on the device."""
    assert ys.dtype == ks.dtype == length.dtype == torch.int32 and ys.is_contiguous() and ks.is_contiguous()
    assert ys.shape[0] >= steps + 1 and ks.shape[0] >= steps and ys[0].numel() == n_sent * bm == ks[0].numel()
    out = torch.zeros((n_sent, n_best, max(steps, 1)), dtype=torch.int32, device=ys.device)
    rc = load().skhip_beam_backtrack(_ptr(ys), _ptr(ks), _ptr(length), _ptr(out), n_sent, bm, n_best, steps, out.shape[2],
                                     _stream())
    _check(rc, "skhip_beam_backtrack")
    return out


def decode_stack_ws_bytes(R):
    return int(load().skhip_decode_stack_ws_bytes(R))


STACK_ERR_WORD = 64  # int32 index of the error word in the stack workspace


def decode_stack_table(layer_tensors):
    """Host table for decode_stack: per layer 22 device tensors in the order of the header."""
    n = len(layer_tensors)
    table = (c_i64 * (22 * n))()
    for i, row in enumerate(layer_tensors):
        assert len(row) == 22
        table[22 * i:22 * i + 22] = [t.data_ptr() for t in row]
    return table, n


def decode_stack(table, n_layers, tok, pos, word_emb, pos_tab, cur_len, row_pad, pad_hist, hist, src_key_pad, dec_out,
                 ws, R, bm, Tc, Ls, d_model, n_head, d_head, d_inner, scale, eps, parity, out_layout=0):
    """One decode step of the whole decoder stack (see the header); ws: uint8 workspace tensor.  out_layout
    ROWS_FRAGMENT_MAJOR: dec_out ([ceil(R / 16) * 16][d_model]) receives the rows as fragment_major_rows orders them."""
    assert tok.dtype == pos.dtype == torch.int64 and cur_len.dtype == hist.dtype == torch.int32
    assert dec_out.dtype == torch.bfloat16 and word_emb.dtype == pos_tab.dtype == torch.float32
    assert dec_out.is_contiguous() and dec_out.numel() >= (R if out_layout == 0 else (R + 15) // 16 * 16) * d_model
    rc = load().skhip_decode_stack(table, n_layers, _ptr(tok), _ptr(pos), _ptr(word_emb), _ptr(pos_tab), _ptr(cur_len),
                                   _ptr(row_pad), _ptr(pad_hist), _ptr(hist), _ptr(src_key_pad), _ptr(dec_out), _ptr(ws),
                                   ws.numel() * ws.element_size(), R, bm, Tc, Ls, d_model, n_head, d_head, d_inner,
                                   float(scale), float(eps), int(parity), int(out_layout), _stream())
    _check(rc, "skhip_decode_stack")


def gather_rows(src, dst, src_row, n_rows, row_elems, src_stride, dst_stride):
    lib = load()
    rc = lib.skhip_gather_rows(_ptr(src), _ptr(dst), _ptr(src_row), n_rows, row_elems, src_stride, dst_stride,
                               dcode(src.dtype), _stream())
    _check(rc, "skhip_gather_rows")
