"""Kernel-level parity: every C-ABI entry point against a plain fp64/fp32 CPU computation of
the same op (torch / the numpy oracle) and against the golden vectors.  GPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import transformer_np as O

pytestmark = pytest.mark.gpu

DEV = "cuda"
F32_RTOL = 1e-5   # north-star tolerance for the fp32 path
BF16_RTOL = 2e-2  # north-star tolerance for bf16 storage


@pytest.fixture(scope="module")
def L():
    from specialk_amd import _lib
    _lib.load()
    return _lib


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def tol(dtype):
    return F32_RTOL if dtype == torch.float32 else BF16_RTOL


def rnd(shape, dtype, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dtype)


# ------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(128, 128, 64), (200, 136, 72), (33, 17, 40), (1, 5, 8), (257, 512, 1000), (130, 70, 37)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_layouts(L, dtype, layout):
    g = torch.Generator().manual_seed(layout * 7 + 1)
    for (M, N, K) in GEMM_SHAPES:
        A = rnd((K, M) if layout == 2 else (M, K), dtype, g)
        B = rnd((N, K) if layout == 0 else (K, N), dtype, g)
        ref = (A.double().t() if layout == 2 else A.double()) @ (B.double().t() if layout == 0 else B.double())
        out = L.gemm(layout, A.to(DEV), B.to(DEV), M, N, K, A.shape[1], B.shape[1], split_k=False)
        assert rel_err(out, ref) <= tol(dtype), (M, N, K)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(L, dtype):
    g = torch.Generator().manual_seed(5)
    M, N, K = 150, 200, 96
    A, B = rnd((M, K), dtype, g), rnd((N, K), dtype, g)
    bias = torch.randn(N, generator=g)
    aux = rnd((M, N), dtype, g)
    base = A.double() @ B.double().t()
    Ad, Bd = A.to(DEV), B.to(DEV)
    out = L.gemm(L.GEMM_NT, Ad, Bd, M, N, K, K, K, bias=bias.to(DEV), epilogue=L.EPI_RELU, alpha=0.5)
    assert rel_err(out, torch.relu(0.5 * base + bias.double())) <= tol(dtype)
    out = L.gemm(L.GEMM_NT, Ad, Bd, M, N, K, K, K, aux=aux.to(DEV), epilogue=L.EPI_RELU_MASK)
    assert rel_err(out, base * (aux.double() > 0)) <= tol(dtype)
    # accumulate into an fp32 C (weight-gradient style) and strided output
    C = torch.randn(M, N + 8, generator=g)
    Cd = C.to(DEV)
    L.gemm(L.GEMM_NT, Ad, Bd, M, N, K, K, K, out=Cd, ldc=N + 8, epilogue=L.EPI_ACCUM)
    assert rel_err(Cd[:, :N], base + C[:, :N].double()) <= tol(dtype)
    assert torch.equal(Cd[:, N:].cpu(), C[:, N:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_splitk_wgrad(L, dtype):
    """Weight-gradient shape: tiny output, long contraction -> deterministic split-K."""
    g = torch.Generator().manual_seed(9)
    T, N, K = 5000, 96, 160
    dY, X = rnd((T, N), dtype, g), rnd((T, K), dtype, g)
    ref = dY.double().t() @ X.double()
    outs = [L.gemm(L.GEMM_TN, dY.to(DEV), X.to(DEV), N, K, T, N, K, out_dtype=torch.float32) for _ in range(2)]
    assert rel_err(outs[0], ref) <= (F32_RTOL if dtype == torch.float32 else 2e-3)
    assert torch.equal(outs[0], outs[1]), "split-K must be bitwise reproducible"


@pytest.mark.parametrize("env", [{"SKHIP_GEMM_BIG": "1"}, {"SKHIP_GEMM_BIG": "0", "SKHIP_GEMM_STAGED": "0"},
                                 {"SKHIP_GEMM_BIG": "0", "SKHIP_GEMM_STAGED": "1"}])
def test_gemm_bf16_kernel_variants(L, env):
    """Every bf16 kernel variant (256x256 direct-to-LDS, 128x128 direct-to-LDS, 128x128 register
    staged), forced through the library's environment switches, on ragged NT shapes with an
    epilogue, a K tail and split-K."""
    import os
    g = torch.Generator().manual_seed(11)
    old = {k: os.environ.get(k) for k in ("SKHIP_GEMM_BIG", "SKHIP_GEMM_STAGED")}
    os.environ.update(env)
    try:
        # the last shape has more 256-tiles (18 x 17) than the chip has CUs
        for (M, N, K) in [(256, 256, 32), (700, 520, 200), (1000, 264, 1096), (513, 8, 72), (3, 1000, 8), (4600, 4100, 104)]:
            A, B = rnd((M, K), torch.bfloat16, g), rnd((N, K), torch.bfloat16, g)
            bias = torch.randn(N, generator=g)
            ref = torch.relu(A.double() @ B.double().t() + bias.double())
            out = L.gemm(L.GEMM_NT, A.to(DEV), B.to(DEV), M, N, K, K, K, bias=bias.to(DEV), epilogue=L.EPI_RELU)
            assert rel_err(out, ref) <= BF16_RTOL, (env, M, N, K)
            out32 = L.gemm(L.GEMM_NT, A.to(DEV), B.to(DEV), M, N, K, K, K, out_dtype=torch.float32, split_k=False)
            assert rel_err(out32, A.double() @ B.double().t()) <= 1e-5, (env, M, N, K)
        # row-contiguous operands (input-gradient NN and weight-gradient TN forms), with and without split-K
        for (M, N, K) in [(256, 256, 64), (520, 264, 1096), (1000, 72, 40), (8, 512, 3000)]:
            A, B = rnd((K, M), torch.bfloat16, g), rnd((K, N), torch.bfloat16, g)
            ref = A.double().t() @ B.double()
            for split in (False, True):
                out = L.gemm(L.GEMM_TN, A.to(DEV), B.to(DEV), M, N, K, M, N, out_dtype=torch.float32, split_k=split)
                assert rel_err(out, ref) <= 1e-5, (env, "TN", M, N, K, split)
            An = A.t().contiguous()
            out = L.gemm(L.GEMM_NN, An.to(DEV), B.to(DEV), M, N, K, K, N, split_k=False)
            assert rel_err(out, ref) <= BF16_RTOL, (env, "NN", M, N, K)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def test_gemm_relu_sign_pattern_epilogues(L):
    """ReLU forward that also writes one sign bit per element, and the masked backward that reads the
    bits: same results as the forms that use the activation itself as the mask; 128- and 256-tile
    kernels."""
    import os
    g = torch.Generator(device=DEV).manual_seed(23)
    old = os.environ.get("SKHIP_GEMM_BIG")
    try:
        for big in ("0", "1"):
            os.environ["SKHIP_GEMM_BIG"] = big
            for (M, N, K) in [(300, 320, 72), (1000, 2048, 512)]:
                x = torch.randn((M, K), device=DEV, generator=g).to(torch.bfloat16)
                W = torch.randn((N, K), device=DEV, generator=g).to(torch.bfloat16)
                b = torch.randn(N, device=DEV, generator=g)
                signs = torch.zeros((M, N // 16), dtype=torch.int16, device=DEV)
                h = L.gemm(L.GEMM_NT, x, W, M, N, K, K, K, bias=b, aux=signs, epilogue=L.EPI_RELU | L.EPI_RELU_BITS_OUT)
                h_ref = L.gemm(L.GEMM_NT, x, W, M, N, K, K, K, bias=b, epilogue=L.EPI_RELU, split_k=False)  # the pattern forms never split K
                assert torch.equal(h, h_ref)
                bits = (signs.to(torch.int32) & 0xFFFF).unsqueeze(-1) >> torch.arange(16, device=DEV) & 1
                assert torch.equal(bits.reshape(M, N).bool(), h_ref > 0)
                dy = torch.randn((M, K), device=DEV, generator=g).to(torch.bfloat16)  # any NT product of this shape
                got = L.gemm(L.GEMM_NT, dy, W, M, N, K, K, K, aux=signs, epilogue=L.EPI_RELU_MASK_BITS)
                ref = L.gemm(L.GEMM_NT, dy, W, M, N, K, K, K, aux=h_ref, epilogue=L.EPI_RELU_MASK, split_k=False)
                assert torch.equal(got, ref)
        with pytest.raises(RuntimeError):  # N not a multiple of 64
            L.gemm(L.GEMM_NT, x, W, 8, 24, K, K, K, aux=signs, epilogue=L.EPI_RELU | L.EPI_RELU_BITS_OUT)
    finally:
        os.environ.pop("SKHIP_GEMM_BIG", None) if old is None else os.environ.__setitem__("SKHIP_GEMM_BIG", old)


def test_gemm_bf16_row_cut(L):
    """Long-K products whose 256-tiles overflow the CUs by a few tiles are computed as two row
    ranges (the second with split-K); operands, output and the ReLU-mask operand must all be
    offset consistently."""
    g = torch.Generator(device=DEV).manual_seed(5)
    M, N, K = 33300, 512, 8200  # 131 x 2 tiles = 256 + 6
    A = torch.randn((K, M), device=DEV, generator=g).to(torch.bfloat16)
    B = torch.randn((K, N), device=DEV, generator=g).to(torch.bfloat16)
    out = L.gemm(L.GEMM_TN, A, B, M, N, K, M, N, out_dtype=torch.float32)
    ref = A.double().t() @ B.double()
    assert rel_err(out.cpu(), ref.cpu()) <= 1e-5
    An = A.t().contiguous()
    Bn = B.t().contiguous()
    aux = torch.randn((M, N), device=DEV, generator=g).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    out = L.gemm(L.GEMM_NT, An, Bn, M, N, K, K, K, bias=bias, aux=aux, epilogue=L.EPI_RELU_MASK)
    ref = torch.where(aux.double() > 0, ref + bias.double(), torch.zeros_like(ref))
    assert rel_err(out.cpu(), ref.cpu()) <= BF16_RTOL


def test_gemm_bf16_tile_heights(L):
    """The 256-tile kernel on lower tiles (tall products whose 256-row tiles would leave the chip's last round
    partly empty): every height gives bit-for-bit the result of the 256-row tiles — plain, with the fused
    epilogues, fp32 output, the NN form — and the plain product agrees with a float64 reference.  M is ragged
    so the last tile of every height is cut by the matrix edge."""
    import os
    g = torch.Generator(device=DEV).manual_seed(41)
    M, N, K = 25613, 512, 136
    x = torch.randn((M, K), device=DEV, generator=g).to(torch.bfloat16)
    W = torch.randn((N, K), device=DEV, generator=g).to(torch.bfloat16)
    b = torch.randn(N, device=DEV, generator=g)
    c0 = torch.randn((M, N), device=DEV, generator=g).to(torch.bfloat16)
    Kn = 8200  # NN products take the 256-tile kernel from K = 8192
    xn = torch.randn((M, Kn), device=DEV, generator=g).to(torch.bfloat16)
    Wn = torch.randn((Kn, N), device=DEV, generator=g).to(torch.bfloat16)

    def run():
        out = {}
        out["plain"] = L.gemm(L.GEMM_NT, x, W, M, N, K, K, K, split_k=False)
        out["f32"] = L.gemm(L.GEMM_NT, x, W, M, N, K, K, K, out_dtype=torch.float32, split_k=False)
        signs = torch.zeros((M, N // 16), dtype=torch.int16, device=DEV)
        out["relu"] = L.gemm(L.GEMM_NT, x, W, M, N, K, K, K, bias=b, aux=signs, alpha=0.5,
                             epilogue=L.EPI_RELU | L.EPI_RELU_BITS_OUT)
        out["signs"] = signs
        out["masked"] = L.gemm(L.GEMM_NT, x, W, M, N, K, K, K, aux=signs, epilogue=L.EPI_RELU_MASK_BITS)
        acc = c0.clone()
        L.gemm(L.GEMM_NT, x, W, M, N, K, K, K, out=acc, ldc=N, epilogue=L.EPI_ACCUM)
        out["accum"] = acc
        out["nn"] = L.gemm(L.GEMM_NN, xn, Wn, M, N, Kn, Kn, N, split_k=False)
        return out

    old = {k: os.environ.get(k) for k in ("SKHIP_GEMM_TILE_H", "SKHIP_GEMM_BIG")}
    try:
        os.environ["SKHIP_GEMM_BIG"] = "1"
        os.environ["SKHIP_GEMM_TILE_H"] = "256"
        ref = run()
        rows = torch.arange(0, M, 97, device=DEV)
        want = x[rows].double() @ W.double().t()
        assert rel_err(ref["plain"][rows], want) <= BF16_RTOL
        assert rel_err(ref["f32"][rows], want) <= 1e-5
        assert rel_err(ref["nn"][rows], xn[rows].double() @ Wn.double()) <= BF16_RTOL
        for th in (144, 160, 176, 192, 208, 224, 240, None):  # None: the library's own choice
            os.environ.pop("SKHIP_GEMM_TILE_H", None) if th is None else os.environ.__setitem__("SKHIP_GEMM_TILE_H", str(th))
            got = run()
            for k in ref:
                assert torch.equal(got[k], ref[k]), (th, k)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


@pytest.mark.parametrize("case", ["split_all", "full_round_and_tail", "many_problems"])
def test_gemm_grouped_tn(L, case):
    """Grouped weight-gradient products: fewer tiles than CUs (everything K-split), a full round of
    tiles plus a K-split remainder cut inside a problem, and more problems than one launch's
    argument table holds; strided operand slices and accumulation into C."""
    g = torch.Generator(device=DEV).manual_seed(17)
    rn = lambda *shape: torch.randn(shape, device=DEV, generator=g).to(torch.bfloat16)
    probs, refs = [], []

    def add(M, N, K, acc, a_pad=0, colsum=None):
        """colsum: None (no column sums wanted), False (written), True (accumulated into)."""
        A_full, B = rn(K, M + a_pad), rn(K, N)
        A = A_full[:, a_pad:]  # a column slice of a wider buffer, like dq|dk|dv
        C = torch.randn((M, N), device=DEV, generator=g) if acc else torch.full((M, N), float("nan"), device=DEV)
        refs.append((C, (C.double() if acc else 0) + A.double().t() @ B.double()))
        cs = None
        if colsum is not None:
            cs = torch.randn(M, device=DEV, generator=g) if colsum else torch.full((M,), float("nan"), device=DEV)
            refs.append((cs, (cs.double() if colsum else 0) + A.double().sum(0)))
        probs.append((A, B, C, M, N, K, A_full.stride(0), N, N, acc, cs, bool(colsum)))

    if case == "split_all":
        add(512, 512, 3000, False, colsum=False); add(512, 512, 3000, True, a_pad=1024, colsum=True)
        add(2048, 512, 3000, True); add(8, 264, 1000, False, colsum=False); add(520, 2048, 2999, False, a_pad=8, colsum=True)
    elif case == "full_round_and_tail":
        for i in range(20):
            add(2048, 512, 1100, i % 2 == 0, colsum=(None, False, True)[i % 3])
        add(1000, 264, 1100, False, colsum=False)
    else:
        for i in range(60):
            add(256 + 8 * (i % 3), 264, 200 + i, i % 3 == 0, colsum=(False, True, None)[i % 3])
    L.gemm_grouped_tn(probs)
    for i, (C, ref) in enumerate(refs):
        assert rel_err(C, ref) <= 1e-5, (case, i)


def test_colsum(L):
    g = torch.Generator().manual_seed(3)
    for dtype in (torch.float32, torch.bfloat16):
        X = rnd((777, 130), dtype, g)
        out = L.colsum(X.to(DEV), 777, 130, 130)
        assert rel_err(out, X.double().sum(0)) <= 1e-5


# ------------------------------------------------------------------------------ attention
def run_attn(L, q, k, v, keypad, causal, dtype, dense=None, p_drop=0.0, seed=0, want_p=False):
    """q (B,Lq,H,dk) etc. as torch CPU tensors; returns O (B,Lq,H,dv), LSE and a closure for bwd."""
    B, Lq, H, dk = q.shape
    Lk, dv = k.shape[1], v.shape[3]
    qd, kd, vd = (t.to(dtype).to(DEV).contiguous() for t in (q, k, v))
    o = torch.empty((B, Lq, H, dv), dtype=dtype, device=DEV)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=DEV)
    kp = keypad.to(torch.uint8).to(DEV) if keypad is not None else None
    dm = dense.to(torch.uint8).to(DEV) if dense is not None else None
    P = torch.empty((H, B, Lq, Lk), dtype=dtype, device=DEV) if want_p else None
    scale = 1.0 / np.sqrt(dk)
    L.attn_fwd(qd, kd, vd, o, lse, B, H, Lq, Lk, dk, dv, H * dk, H * dk, H * dv, H * dv, scale, key_pad=kp,
               dense_mask=dm, causal=causal, p_drop=p_drop, seed=seed, P_out=P)

    def bwd(do):
        dod = do.to(dtype).to(DEV).contiguous()
        dq, dk_, dv_ = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        delta = torch.empty((B, H, Lq), dtype=torch.float32, device=DEV)
        L.attn_bwd(qd, kd, vd, o, dod, lse, dq, dk_, dv_, delta, B, H, Lq, Lk, dk, dv, H * dk, H * dk, H * dv, H * dv,
                   H * dk, H * dk, H * dv, scale, key_pad=kp, dense_mask=dm, causal=causal, p_drop=p_drop, seed=seed)
        return dq, dk_, dv_

    return o, lse, P, bwd


def oracle_attn(q, k, v, keypad, causal, dense=None):
    """fp64 oracle on (B,L,H,d) tensors -> same layout."""
    B, Lq, H, dk = q.shape
    Lk = k.shape[1]
    hm = lambda t: t.double().permute(2, 0, 1, 3).reshape(H * B, t.shape[1], t.shape[3]).numpy()
    mask = np.zeros((B, Lq, Lk), dtype=bool)
    if keypad is not None:
        mask |= keypad.numpy().astype(bool)[:, None, :]
    if causal:
        mask |= np.triu(np.ones((Lq, Lk), dtype=bool), k=1)[None]
    if dense is not None:
        mask |= dense.numpy().astype(bool)
    mask_h = np.tile(mask, (H, 1, 1))
    qh, kh, vh = hm(q), hm(k), hm(v)
    out, p = O.sdpa_fwd(qh, kh, vh, mask_h, np.sqrt(dk))
    back = lambda a: torch.from_numpy(a.reshape(H, B, a.shape[1], a.shape[2])).permute(1, 2, 0, 3).contiguous()

    def bwd(do):
        dq, dk_, dv_ = O.sdpa_bwd(qh, kh, vh, p, hm(do), np.sqrt(dk))
        return back(dq), back(dk_), back(dv_)

    return back(out), torch.from_numpy(p.reshape(H, B, Lq, Lk)), bwd


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["self_pad", "causal", "cross", "long", "dense", "d128", "h16_causal", "h16_cross"])
def test_attention(L, dtype, case):
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    B, H, Lq, Lk, dk, dv, causal, dense = {
        "self_pad": (3, 4, 21, 21, 32, 32, False, False),
        "causal": (2, 8, 50, 50, 64, 64, True, False),
        "cross": (3, 2, 17, 70, 64, 64, False, False),
        "long": (1, 2, 150, 200, 64, 64, False, False),
        "dense": (2, 2, 9, 11, 16, 24, False, True),
        "d128": (2, 2, 20, 33, 128, 128, False, False),
        "h16_causal": (3, 16, 52, 52, 64, 64, True, False),   # the Transformer-big head count (d = 16 x 64)
        "h16_cross": (3, 16, 41, 52, 64, 64, False, False),
    }[case]
    q, k, v = rnd((B, Lq, H, dk), dtype, g), rnd((B, Lk, H, dk), dtype, g), rnd((B, Lk, H, dv), dtype, g)
    lens = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
    lens[0] = Lk
    keypad = torch.arange(Lk)[None, :] >= lens[:, None]
    dm = None
    if dense:
        dm = torch.rand(B, Lq, Lk, generator=g) < 0.3
        dm[:, :, 0] = False
    o, lse, P, bwd = run_attn(L, q, k, v, keypad, causal, dtype, dense=dm, want_p=True)
    ro, rp, rbwd = oracle_attn(q, k, v, keypad, causal, dense=dm)
    t = tol(dtype)
    assert rel_err(o, ro) <= t
    assert rel_err(P, rp) <= t
    do = rnd(o.shape, dtype, g)
    for got, ref in zip(bwd(do), rbwd(do)):
        assert rel_err(got, ref) <= t
    if dtype == torch.bfloat16 and not dense:
        # without the probability output the bf16 call takes the MFMA kernels (dk == dv in {32,64,128})
        o2, lse2, _, bwd2 = run_attn(L, q, k, v, keypad, causal, dtype)
        assert rel_err(o2, ro) <= t
        assert rel_err(lse2, lse) <= 1e-2
        for got, ref in zip(bwd2(do), rbwd(do)):
            assert rel_err(got, ref) <= t


@pytest.mark.parametrize("Lq,Lk", [(70, 90), (40, 52)])  # two-launch backward / fused single-tile backward
def test_attention_mfma_dropout_matches_valu(L, Lq, Lk):
    """Same seed -> the MFMA and the vector-ALU kernels drop the same probabilities, forward and backward."""
    import os
    g = torch.Generator().manual_seed(5)
    B, H, d = 3, 4, 64
    dt = torch.bfloat16
    q, k, v = rnd((B, Lq, H, d), dt, g), rnd((B, Lk, H, d), dt, g), rnd((B, Lk, H, d), dt, g)
    lens = torch.tensor([Lk, Lk // 2 + 5, Lk - 13])
    keypad = torch.arange(Lk)[None, :] >= lens[:, None]
    do = rnd((B, Lq, H, d), dt, g)
    res = {}
    for mode in ("0", "1"):
        os.environ["SKHIP_ATTN_VALU"] = mode
        try:
            o, lse, _, bwd = run_attn(L, q, k, v, keypad, False, dt, p_drop=0.1, seed=99)
            res[mode] = (o, lse) + tuple(bwd(do))
        finally:
            os.environ.pop("SKHIP_ATTN_VALU", None)
    for a, b in zip(res["0"], res["1"]):
        assert rel_err(a, b) <= 2e-2


def test_attention_golden(L):
    gd = load_golden("attention")
    H = int(gd["n_head"])
    for tag, causal in (("keypad", False), ("causal", True), ("cross", False)):
        unhead = lambda a: torch.from_numpy(a).reshape(H, -1, a.shape[1], a.shape[2]).permute(1, 2, 0, 3).contiguous()
        q, k, v, do = (unhead(gd[f"{tag}_{n}"]) for n in ("q", "k", "v", "dout"))
        keypad = torch.from_numpy(gd[f"{tag}_keypad"])
        o, lse, P, bwd = run_attn(L, q, k, v, keypad, causal, torch.float32, want_p=True)
        assert rel_err(o, unhead(gd[f"{tag}_out"])) <= F32_RTOL
        assert rel_err(P, torch.from_numpy(gd[f"{tag}_attn"]).reshape(P.shape)) <= F32_RTOL
        for got, name in zip(bwd(do), ("dq", "dk", "dv")):
            assert rel_err(got, unhead(gd[f"{tag}_{name}"])) <= F32_RTOL


def test_attention_fully_masked_row_is_nan(L):
    q = torch.randn(1, 3, 1, 16); k = torch.randn(1, 4, 1, 16); v = torch.randn(1, 4, 1, 16)
    keypad = torch.tensor([[True, True, True, True]])
    o, _, _, _ = run_attn(L, q, k, v, keypad, False, torch.float32)
    assert torch.isnan(o).all()


def test_attention_dropout_statistics(L):
    g = torch.Generator().manual_seed(77)
    B, H, Lq, Lk, d = 4, 4, 48, 48, 32
    q, k, v = rnd((B, Lq, H, d), torch.float32, g), rnd((B, Lk, H, d), torch.float32, g), rnd((B, Lk, H, d), torch.float32, g)
    _, _, P0, _ = run_attn(L, q, k, v, None, False, torch.float32, want_p=True)
    o1, _, P1, bwd1 = run_attn(L, q, k, v, None, False, torch.float32, p_drop=0.1, seed=123, want_p=True)
    o2, _, P2, _ = run_attn(L, q, k, v, None, False, torch.float32, p_drop=0.1, seed=123, want_p=True)
    assert torch.equal(P1, P2) and torch.equal(o1, o2), "same seed -> same mask"
    kept = P1 != 0
    rate = kept.float().mean().item()
    assert abs(rate - 0.9) < 0.01, rate
    assert rel_err(P1[kept], (P0 / 0.9)[kept]) <= 1e-5
    # O must equal P_dropped @ V, and backward must use the same mask
    Ph = P1.double().cpu()                                   # (H,B,Lq,Lk)
    vh = v.double().permute(2, 0, 1, 3)                      # (H,B,Lk,d)
    assert rel_err(o1, (Ph @ vh).permute(1, 2, 0, 3)) <= 1e-5
    do = rnd(o1.shape, torch.float32, g)
    dq, dk_, dv_ = bwd1(do)
    assert rel_err(dv_, (Ph.transpose(2, 3) @ do.double().permute(2, 0, 1, 3)).permute(1, 2, 0, 3)) <= 1e-5
    # dQ via autograd on the same mask
    qa = q.double().permute(2, 0, 1, 3).requires_grad_(True)
    ka = k.double().permute(2, 0, 1, 3).requires_grad_(True)
    S = torch.softmax(qa @ ka.transpose(2, 3) / np.sqrt(d), -1) * (kept.cpu().double() / 0.9)
    ((S @ vh) * do.double().permute(2, 0, 1, 3)).sum().backward()
    assert rel_err(dq, qa.grad.permute(1, 2, 0, 3)) <= 1e-5
    assert rel_err(dk_, ka.grad.permute(1, 2, 0, 3)) <= 1e-5


# ------------------------------------------------------------------------------ add + LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d", [32, 256, 512, 1024])
def test_add_ln(L, dtype, d):
    g = torch.Generator().manual_seed(d)
    T = 203
    x, res = rnd((T, d), dtype, g), rnd((T, d), dtype, g)
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    pad = torch.rand(T, generator=g) < 0.2
    y, mean, rstd = L.add_ln_fwd(x.to(DEV), res.to(DEV), gamma.to(DEV), beta.to(DEV), pad.to(torch.uint8).to(DEV), 1e-5, 0.0, 0)
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr + rr, (d,), gr, br, 1e-5) * (~pad)[:, None]
    assert rel_err(y, ref) <= tol(dtype)
    assert (y[pad.to(DEV)] == 0).all()
    dy = rnd((T, d), dtype, g)
    ref.backward(dy.double())
    dx, dres, dg, db = L.add_ln_bwd(dy.to(DEV), x.to(DEV), res.to(DEV), gamma.to(DEV), pad.to(torch.uint8).to(DEV), mean, rstd, 0.0, 0)
    assert dx.data_ptr() == dres.data_ptr()
    assert rel_err(dres, rr.grad) <= tol(dtype)
    assert rel_err(dg, gr.grad) <= max(tol(dtype) / 4, 1e-5)
    assert rel_err(db, br.grad) <= max(tol(dtype) / 4, 1e-5)


def test_add_ln_dropout(L):
    g = torch.Generator().manual_seed(8)
    T, d, p = 512, 256, 0.1
    x, res = rnd((T, d), torch.float32, g), rnd((T, d), torch.float32, g)
    gamma, beta = torch.ones(d), torch.zeros(d)
    xd, rd, gd, bd = x.to(DEV), res.to(DEV), gamma.to(DEV), beta.to(DEV)
    y, mean, rstd = L.add_ln_fwd(xd, rd, gd, bd, None, 1e-5, p, 42)
    dy = rnd((T, d), torch.float32, g).to(DEV)
    dx, dres, _, _ = L.add_ln_bwd(dy, xd, rd, gd, None, mean, rstd, p, 42)
    kept = (dx != 0).cpu()
    assert abs(kept.float().mean().item() - (1 - p)) < 0.01
    assert rel_err(dx.cpu()[kept], (dres.cpu() / (1 - p))[kept]) <= 1e-6
    xr = (x * kept / (1 - p)).double()
    ref = torch.nn.functional.layer_norm(xr + res.double(), (d,), None, None, 1e-5)
    assert rel_err(y, ref) <= 1e-5


# ------------------------------------------------------------------------------ embeddings
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed(L, dtype):
    g = torch.Generator().manual_seed(2)
    V, P, d, T = 50, 12, 64, 37
    W, Pt = torch.randn(V, d, generator=g), torch.randn(P, d, generator=g)
    seq, pos = torch.randint(0, V, (T,), generator=g), torch.randint(0, P, (T,), generator=g)
    seq[:5] = 0
    out = L.embed_fwd(seq.to(DEV), pos.to(DEV), W.to(DEV), Pt.to(DEV), dtype)
    assert rel_err(out, W[seq] + Pt[pos]) <= (1e-7 if dtype == torch.float32 else 1e-2)
    dout = rnd((T, d), dtype, g)
    dW = torch.zeros(V, d, device=DEV)
    L.embed_bwd(seq.to(DEV), dout.to(DEV), dW, 0)
    ref = torch.zeros(V, d, dtype=torch.float64).index_add_(0, seq, dout.double())
    ref[0] = 0
    assert rel_err(dW, ref) <= 1e-6


# ------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V", [23, 4096, 35000])
def test_xent(L, dtype, V):
    g = torch.Generator().manual_seed(V)
    T = 40
    logits = rnd((T, V), dtype, g, 3.0)
    gold = torch.randint(1, V, (T,), generator=g)
    gold[::7] = 0
    for eps, scale in ((0.1, 1.0), (0.0, 1.0), (0.1, 0.125)):
        ld = logits.to(DEV)
        dl = torch.empty_like(ld)
        stats = torch.zeros(3, dtype=torch.float64, device=DEV)
        L.xent_fwd_bwd(ld, gold.to(DEV), dl, stats, scale, eps, 0)
        loss, dpred, nc = O.ls_xent(logits.double().numpy() * scale, gold.numpy(), eps > 0, eps if eps > 0 else 0.1)
        s = stats.cpu().numpy()
        assert abs(s[0] - loss) <= (1e-5 if dtype == torch.float32 else 1e-3) * abs(loss)
        assert int(s[1]) == nc and int(s[2]) == int((gold != 0).sum())
        assert rel_err(dl, torch.from_numpy(dpred * scale)) <= tol(dtype)
    # in place (dlogits aliases logits) gives the same bits as out of place
    ld, ld2 = logits.to(DEV), logits.to(DEV)
    dl = torch.empty_like(ld)
    stats = torch.zeros(3, dtype=torch.float64, device=DEV)
    L.xent_fwd_bwd(ld, gold.to(DEV), dl, stats, 1.0, 0.1, 0)
    L.xent_fwd_bwd(ld2, gold.to(DEV), ld2, stats, 1.0, 0.1, 0)
    assert torch.equal(ld2, dl)


def test_xent_golden(L):
    gd = load_golden("loss")
    pred, gold = torch.from_numpy(gd["pred"]).to(DEV), torch.from_numpy(gd["gold"]).to(DEV)
    for s, eps in (("ls", 0.1), ("ce", 0.0)):
        dl = torch.empty_like(pred)
        stats = torch.zeros(3, dtype=torch.float64, device=DEV)
        L.xent_fwd_bwd(pred, gold, dl, stats, 1.0, eps, 0)
        assert abs(stats[0].item() - float(gd[f"loss_{s}"])) <= 1e-5 * abs(float(gd[f"loss_{s}"]))
        assert int(stats[1].item()) == int(gd[f"ncorrect_{s}"])
        assert rel_err(dl, torch.from_numpy(gd[f"dpred_{s}"])) <= 1e-5


# ------------------------------------------------------------------------------ Adam
def make_tables(tensors_by_key):
    out = {}
    for k, ts in tensors_by_key.items():
        if ts is None:
            out[k] = None
        elif k == "n":
            out[k] = torch.tensor(ts, dtype=torch.int64, device=DEV)
        else:
            out[k] = torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=DEV)
    return out


def test_adam_golden(L):
    gd = load_golden("optim")
    d_model, warm = int(gd["d_model"]), int(gd["n_warmup"])
    p = [torch.from_numpy(gd["p1_0"]).to(DEV), torch.from_numpy(gd["p2_0"]).to(DEV)]
    m = [torch.zeros_like(t) for t in p]
    v = [torch.zeros_like(t) for t in p]
    gbuf = [torch.zeros_like(t) for t in p]
    b = [torch.zeros(t.shape, dtype=torch.bfloat16, device=DEV) for t in p]
    tables = make_tables({"p": p, "g": gbuf, "m": m, "v": v, "b": b, "n": [t.numel() for t in p]})
    for step in range(1, 7):
        gbuf[0].copy_(torch.from_numpy(gd[f"g1_{step}"]))
        gbuf[1].copy_(torch.from_numpy(gd[f"g2_{step}"]))
        lr = O.noam_lr(step, d_model, warm)
        L.adam_multi(tables, 2, max(t.numel() for t in p), lr, 0.9, 0.98, 1e-9, step)
        assert rel_err(p[0], torch.from_numpy(gd[f"p1_{step}"])) <= 1e-6
        assert rel_err(p[1], torch.from_numpy(gd[f"p2_{step}"])) <= 1e-6
        assert torch.equal(b[0], p[0].to(torch.bfloat16)) and torch.equal(b[1], p[1].to(torch.bfloat16))


def test_fused_adam_tables_survive_a_host_that_runs_ahead(L):
    """Ten optimiser steps enqueued behind a long-running kernel, with no host synchronisation between them and a
    freshly allocated gradient tensor per step (so every step uploads a different pointer table): each step
    must read ITS table, not a later one's.  Checked against torch.optim.Adam on the CPU."""
    from specialk_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(21)
    w0 = [torch.randn(300, 40, generator=g), torch.randn(77, generator=g)]
    grads = [[torch.randn(w.shape, generator=g) for w in w0] for _ in range(10)]
    ref_p = [torch.nn.Parameter(w.clone()) for w in w0]
    ref = torch.optim.Adam(ref_p, lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
    for gs in grads:
        for p, gr in zip(ref_p, gs):
            p.grad = gr.clone()
        ref.step()
    dev_p = [torch.nn.Parameter(w.clone().to(DEV)) for w in w0]
    opt = FusedAdam(dev_p, lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
    dev_grads = [[gr.to(DEV) for gr in gs] for gs in grads]  # all alive at once: ten distinct addresses per parameter
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e8))  # ~0.1 s of GPU time: the host enqueues all ten steps meanwhile
    for gs in dev_grads:
        for p, gr in zip(dev_p, gs):
            p.grad = gr
        opt.step()
    torch.cuda.synchronize()
    for p, r in zip(dev_p, ref_p):
        assert rel_err(p, r) <= 1e-5


def test_cast_and_convert(L):
    a = [torch.randn(1000, device=DEV), torch.randn(3, 5, device=DEV)]
    b = [torch.empty(t.shape, dtype=torch.bfloat16, device=DEV) for t in a]
    t = make_tables({"s": a, "d": b, "n": [x.numel() for x in a]})
    L.cast_multi(t["s"], t["d"], t["n"], 2, 1000)
    for x, y in zip(a, b):
        assert torch.equal(y, x.to(torch.bfloat16))
    out = L.convert(b[0], torch.empty(1000, device=DEV))
    assert torch.equal(out, b[0].float())


def test_pack_plan(L):
    """Lengths, offsets, row indices and the host facts of right-padded id batches, with and without a
    trimmed last token; more sentences than the block has threads; a batch that is not right-padded."""
    g = torch.Generator().manual_seed(12)
    for B, Lmax, trim in ((7, 9, 0), (640, 52, 1), (2500, 6, 1), (1, 1, 0)):
        lens = torch.randint(1 if trim == 0 else 2, Lmax + 1, (B,), generator=g) if Lmax > 1 else torch.ones(B, dtype=torch.long)
        seq = torch.zeros(B, Lmax, dtype=torch.long)
        for b in range(B):
            seq[b, :lens[b]] = torch.randint(1, 50, (int(lens[b]),), generator=g)
        ln, off, idx, facts = L.pack_plan(seq.to(DEV), trim, 0)
        want = (lens - trim).clamp_min(0)
        assert torch.equal(ln.cpu().long(), want)
        assert torch.equal(off.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), want.cumsum(0)]))
        n = int(want.sum())
        assert facts.cpu().tolist() == [1, int((lens - trim).min()), int(want.max()), n]
        ref_idx = torch.cat([b * Lmax + torch.arange(int(want[b])) for b in range(B)]) if n else torch.zeros(0, dtype=torch.long)
        assert torch.equal(idx[:n].cpu(), ref_idx)
    bad = torch.tensor([[5, 0, 6, 0], [7, 8, 0, 0]])
    assert L.pack_plan(bad.to(DEV), 0, 0)[3].cpu().tolist()[0] == 0
    short = torch.tensor([[5, 0, 0], [7, 8, 9]])
    assert L.pack_plan(short.to(DEV), 2, 0)[3].cpu().tolist()[:2] == [1, -1]


def test_pack_plans_matches_pack_plan_and_checks_ranges(L):
    """Several batches in one call: the same lengths / offsets / indices / facts as skhip_pack_plan per batch (rows
    longer than a wave, more sentences than the block has waves x 4, strided rows), and the two range flags."""
    g = torch.Generator().manual_seed(13)

    def make(B, Lmax, lo=1):
        lens = torch.randint(lo, Lmax + 1, (B,), generator=g)
        seq = torch.zeros(B, Lmax, dtype=torch.long)
        for b in range(B):
            seq[b, :lens[b]] = torch.randint(1, 50, (int(lens[b]),), generator=g)
        pos = (torch.arange(Lmax)[None, :] + 1) * (seq != 0)
        return seq.to(DEV), pos.to(DEV)

    a, b, c = make(640, 53, 2), make(7, 150), make(2500, 6, 2)
    wide = torch.zeros(640, 60, dtype=torch.long, device=DEV)
    wide[:, :53] = a[0]
    items = [(wide[:, :53], a[1], 1, 50, 54), (b[0], None, 0, 0, 0), (c[0], c[1], 1, 50, 7)]
    outs, facts = L.pack_plans(items, 0)
    facts = facts.cpu().tolist()
    for (seq, pos, trim, nv, npos), (ln, off, idx), f in zip(items, outs, facts):
        ln1, off1, idx1, f1 = L.pack_plan(seq.contiguous(), trim, 0)
        n = int(off1[-1])
        assert torch.equal(ln, ln1) and torch.equal(off, off1) and torch.equal(idx[:n], idx1[:n])
        assert f[:4] == f1.cpu().tolist() and f[4:] == [1, 1]
    # an id equal to the vocabulary size, a negative id, a position equal to the table size: flagged, each by itself
    for which, (r, cidx, val) in {"id": (3, 1, 50), "neg": (600, 0, -1), "pos": (77, 2, 54)}.items():
        seq, pos = a[0].clone(), a[1].clone()
        (pos if which == "pos" else seq)[r, cidx] = val
        _, f = L.pack_plans([(seq, pos, 1, 50, 54)], 0)
        assert f.cpu().tolist()[0][4:] == ([1, 0] if which == "pos" else [0, 1]), which
    # not right-padded; a sentence shorter than the trim; an empty batch beside a full one
    bad = torch.tensor([[5, 0, 6, 0], [7, 8, 0, 0]], device=DEV)
    short = torch.tensor([[5, 0, 0], [7, 8, 9]], device=DEV)
    outs, f = L.pack_plans([(bad, None, 0, 0, 0), (short, None, 2, 0, 0), (torch.zeros(0, 5, dtype=torch.long, device=DEV), None, 0, 0, 0)], 0)
    f = f.cpu().tolist()
    assert f[0][0] == 0 and f[1][:2] == [1, -1] and f[2][:4] == [1, 0x7fffffff, 0, 0]


# ------------------------------------------------------------------------------ decode
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decode_attn(L, dtype):
    g = torch.Generator().manual_seed(4)
    R, H, Lmax, dk, dv = 7, 4, 90, 64, 64
    q = rnd((R, H * dk), dtype, g)
    Kc, Vc = rnd((R, Lmax, H * dk), dtype, g), rnd((R, Lmax, H * dv), dtype, g)
    lens = torch.randint(1, Lmax + 1, (R,), generator=g).to(torch.int32)
    lens[0] = Lmax
    keypad = torch.rand(R, Lmax, generator=g) < 0.2
    keypad[:, 0] = False
    out = torch.empty((R, H * dv), dtype=dtype, device=DEV)
    L.decode_attn(q.to(DEV), Kc.to(DEV), Vc.to(DEV), out, R, H, Lmax, Lmax, dk, dv, H * dk, H * dv, 1 / np.sqrt(dk),
                  kv_len=lens.to(DEV), key_pad=keypad.to(torch.uint8).to(DEV))
    mask = keypad | (torch.arange(Lmax)[None] >= lens[:, None])
    ro, _, _ = oracle_attn(q.view(R, 1, H, dk), Kc.view(R, Lmax, H, dk), Vc.view(R, Lmax, H, dv), mask, False)
    assert rel_err(out.view(R, 1, H, dv), ro) <= tol(dtype)
    # the same histories scattered over the cache rows (a different row permutation per position),
    # found through the history-row table: identical output
    perm = torch.stack([torch.randperm(R, generator=g) for _ in range(Lmax)], 1)           # (R, Lmax): row of (r, j)
    Ks, Vs, pads = torch.zeros_like(Kc), torch.zeros_like(Vc), torch.zeros(R, Lmax, dtype=torch.uint8)
    cols = torch.arange(Lmax)[None].expand(R, Lmax)
    Ks[perm, cols], Vs[perm, cols], pads[perm, cols] = Kc, Vc, keypad.to(torch.uint8)
    out2 = torch.empty_like(out)
    L.decode_attn(q.to(DEV), Ks.to(DEV), Vs.to(DEV), out2, R, H, Lmax, Lmax, dk, dv, H * dk, H * dv, 1 / np.sqrt(dk),
                  kv_len=lens.to(DEV), key_pad=pads.to(DEV), hist_row=perm.to(torch.int32).contiguous().to(DEV))
    assert torch.equal(out2, out)
    # append form: the newest position's key / value come from a projection buffer and are written into
    # the caches by the kernel (common length, no kv_len)
    n = 37
    newkv = torch.cat([rnd((R, H * dk), dtype, g), rnd((R, 24), dtype, g), rnd((R, H * dv), dtype, g)], 1).to(DEV)
    Kd, Vd = Ks.to(DEV).clone(), Vs.to(DEV).clone()
    hist = perm.to(torch.int32).contiguous().to(DEV)
    out3 = torch.empty_like(out)
    L.decode_attn(q.to(DEV), Kd, Vd, out3, R, H, n, Lmax, dk, dv, H * dk, H * dv, 1 / np.sqrt(dk), key_pad=pads.to(DEV),
                  hist_row=hist, new_k=newkv, new_v=newkv[:, H * dk + 24:], ld_new=newkv.shape[1])
    Kref, Vref = Kc.clone(), Vc.clone()
    Kref[:, n - 1], Vref[:, n - 1] = newkv[:, :H * dk].cpu(), newkv[:, H * dk + 24:].cpu()
    m3 = keypad | (torch.arange(Lmax)[None] >= n)
    ro3, _, _ = oracle_attn(q.view(R, 1, H, dk), Kref.view(R, Lmax, H, dk), Vref.view(R, Lmax, H, dv), m3, False)
    assert rel_err(out3.view(R, 1, H, dv), ro3) <= tol(dtype)
    rows = perm[:, n - 1]
    assert torch.equal(Kd[rows, n - 1].cpu(), Kref[:, n - 1]) and torch.equal(Vd[rows, n - 1].cpu(), Vref[:, n - 1])
    untouched = torch.ones(R, Lmax, dtype=torch.bool); untouched[rows, n - 1] = False
    assert torch.equal(Kd.cpu()[untouched], Ks[untouched]) and torch.equal(Vd.cpu()[untouched], Vs[untouched])


@pytest.mark.parametrize("bm", [1, 3, 5, 8])
def test_beam_topk(L, bm):
    g = torch.Generator().manual_seed(bm)
    for V in (24, 4096, 35000):
        n_sent = 6
        logits = torch.randn(n_sent * bm, V, generator=g) * 2
        logits[1, 7] = logits[1, 3]  # an exact tie inside a row
        scores = -torch.rand(n_sent, bm, generator=g) * 5
        for first in (True, False):
            so, prev, tok = L.beam_topk(logits.to(DEV), scores.to(DEV), n_sent, bm, first)
            lp = torch.log_softmax(logits, 1).view(n_sent, bm, V)
            for s in range(n_sent):
                flat = (lp[s, 0] if first else (lp[s] + scores[s][:, None]).reshape(-1)).numpy()
                best, ids = O.stable_topk(flat, bm)
                assert (prev[s].cpu().numpy() == ids // V).all(), (V, s, first)
                assert (tok[s].cpu().numpy() == ids % V).all(), (V, s, first)
                np.testing.assert_allclose(so[s].cpu().numpy(), best, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("V,bm,rows_of", [(35000, 5, 128), (1003, 5, 7), (4096, 3, 11), (200, 16, 2)])
def test_beam_topk_from_generator_statistics(L, V, bm, rows_of):
    """SK_EPI_ROWSTATS + skhip_beam_topk_stats (the bf16 decode step's form) against the plain product +
    skhip_beam_topk: the same logits bit for bit, chunk statistics equal to torch's, the same tokens and
    back-pointers — with duplicated weight rows, i.e. exact ties between tokens of one row, inside a chunk and
    across chunks, and V not a multiple of 64 — and scores to fp32 rounding."""
    g = torch.Generator().manual_seed(V + bm)
    n_sent, d = rows_of, 512
    R = n_sent * bm
    x = rnd((R, d), torch.bfloat16, g).to(DEV)
    W = (rnd((V, d), torch.bfloat16, g) * 0.2)
    W[9] = W[3]            # a tie inside a chunk
    if V > 150:
        W[140] = W[3]      # ... and across chunks
        W[V - 1] = W[V - 2]
    W = W.to(DEV)
    scores = (-torch.rand(n_sent, bm, generator=g) * 5).to(DEV)
    plain = L.gemm(L.GEMM_NT, x, W, R, V, d, d, d, out_dtype=torch.float32, split_k=False)
    logits = torch.empty((R, V), dtype=torch.float32, device=DEV)
    nch = (V + 63) // 64
    stats = torch.full((R, nch, 2), float("nan"), dtype=torch.float32, device=DEV)
    L.gemm(L.GEMM_NT, x, W, R, V, d, d, d, out=logits, ldc=V, aux=stats, epilogue=L.EPI_ROWSTATS, split_k=False)
    assert torch.equal(logits, plain)
    pad = torch.full((R, nch * 64), float("-inf"), dtype=torch.float64, device=DEV)
    pad[:, :V] = logits.double()
    ch = pad.view(R, nch, 64)
    m = ch.max(2).values
    se = torch.exp(ch - m[:, :, None]).sum(2)
    assert torch.equal(stats[:, :, 0].double(), m)
    assert rel_err(stats[:, :, 1], se) <= 1e-5
    for first in (True, False):
        so, prev, tok = L.beam_topk(logits, scores, n_sent, bm, first)
        out = (torch.empty_like(so), torch.empty_like(prev), torch.empty_like(tok))
        so2, prev2, tok2 = L.beam_topk_stats(logits, stats, scores, n_sent, bm, first, out)
        assert torch.equal(prev2, prev) and torch.equal(tok2, tok), (V, bm, first)
        assert (so2 - so).abs().max().item() <= 2e-5


@pytest.mark.parametrize("V,bm,n_sent", [(35000, 5, 128), (1003, 5, 7), (200, 16, 2), (4096, 1, 3)])
def test_beam_step_equals_topk_then_update(L, V, bm, n_sent):
    """skhip_beam_step against skhip_beam_topk_stats + skhip_beam_update on the same state: every output and every
    piece of state bit for bit, at the first step (beam 0 only) and at a later one with finished sentences, EOS on
    a best beam, padding tokens and re-parented histories."""
    g = torch.Generator().manual_seed(31 * V + bm)
    R, d, Tc, Tmax, PAD, EOS = n_sent * bm, 512, 12, 11, 0, 3
    x = rnd((R, d), torch.bfloat16, g).to(DEV)
    W = (rnd((V, d), torch.bfloat16, g) * 0.2).to(DEV)
    logits = torch.empty((R, V), dtype=torch.float32, device=DEV)
    stats = torch.empty((R, (V + 63) // 64, 2), dtype=torch.float32, device=DEV)
    L.gemm(L.GEMM_NT, x, W, R, V, d, d, d, out=logits, ldc=V, aux=stats, epilogue=L.EPI_ROWSTATS, split_k=False)
    logits[0, EOS] = 50.0   # sentence 0 ends at this step
    if R > bm:
        logits[bm, PAD] = 50.0   # a padding token on a best beam
    # (statistics refreshed for the edited rows)
    pad = torch.full((R, stats.shape[1] * 64), float("-inf"), dtype=torch.float32, device=DEV)
    pad[:, :V] = logits
    ch = pad.view(R, -1, 64)
    stats[:, :, 0] = ch.max(2).values
    stats[:, :, 1] = torch.exp(ch - stats[:, :, 0:1]).sum(2)
    i32 = dict(dtype=torch.int32, device=DEV)
    for first, t in ((True, 1), (False, 6)):
        def state():
            gs = torch.Generator().manual_seed(t)
            st = dict(scores=(-torch.rand(n_sent, bm, generator=gs) * 5).to(DEV), ys=torch.zeros((Tmax + 1, n_sent, bm), **i32),
                      ks=torch.zeros((Tmax, n_sent, bm), **i32), done=torch.zeros(n_sent, **i32),
                      length=torch.full((n_sent,), Tmax, **i32), n_done=torch.zeros(1, **i32),
                      cur_len=torch.full((R,), t, **i32), tok=torch.zeros(R, dtype=torch.int64, device=DEV),
                      pos=torch.zeros(R, dtype=torch.int64, device=DEV), row_pad=torch.zeros(R, dtype=torch.uint8, device=DEV),
                      pad_hist=torch.zeros((R, Tc), dtype=torch.uint8, device=DEV),
                      hist_in=torch.randint(0, R, (R, Tc), generator=gs, dtype=torch.int32).to(DEV),
                      hist_out=torch.full((R, Tc), -1, **i32))
            st["scores"][0, 1:] = -30.0   # sentence 0's best candidate comes from beam 0, whose best token is EOS
            if n_sent > 2:
                st["done"][2] = 1   # a sentence that finished earlier keeps its scores
                st["n_done"][0] = 1
            return st
        a, b = state(), state()
        outs = [tuple(torch.empty((n_sent, bm), dtype=dt, device=DEV) for dt in (torch.float32, torch.int32, torch.int32))
                for _ in range(2)]
        L.beam_topk_stats(logits, stats, a["scores"], n_sent, bm, first, outs[0])
        L.beam_update(*outs[0], a["scores"], a["ys"], a["ks"], a["done"], a["length"], a["n_done"], a["cur_len"], a["tok"],
                      a["pos"], a["row_pad"], a["pad_hist"], a["hist_in"], a["hist_out"], n_sent, bm, Tc, Tmax, PAD, EOS)
        L.beam_step(logits, stats, outs[1], b["scores"], b["ys"], b["ks"], b["done"], b["length"], b["n_done"], b["cur_len"],
                    b["tok"], b["pos"], b["row_pad"], b["pad_hist"], b["hist_in"], b["hist_out"], n_sent, bm, first, Tc, Tmax,
                    PAD, EOS)
        for u, v in zip(outs[0], outs[1]):
            assert torch.equal(u, v), (V, bm, first)
        for k in a:
            assert torch.equal(a[k], b[k]), (k, V, bm, first)
        assert a["done"][0].item() == 1 and a["length"][0].item() == t   # the EOS case was exercised


def test_gather_rows(L):
    g = torch.Generator().manual_seed(6)
    for dtype, elems in ((torch.float32, 96), (torch.bfloat16, 100), (torch.bfloat16, 64)):
        src = rnd((9, 128), dtype, g).to(DEV)
        idx = torch.tensor([3, 3, 0, 8, 1], dtype=torch.int32, device=DEV)
        dst = torch.zeros((5, 128), dtype=dtype, device=DEV)
        L.gather_rows(src, dst, idx, 5, elems, 128, 128)
        assert torch.equal(dst[:, :elems], src[idx.long(), :elems])
        assert (dst[:, elems:] == 0).all()


def test_add_ln_bwd_dx_colsum(L):
    """The LayerNorm backward can emit sum_rows(dx): the bias gradient of the linear that produced x."""
    g = torch.Generator().manual_seed(12)
    T, d = 777, 512
    for dtype, p in ((torch.float32, 0.0), (torch.bfloat16, 0.1)):
        x, res, dy = (rnd((T, d), dtype, g).to(DEV) for _ in range(3))
        gamma, beta = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
        _, mean, rstd = L.add_ln_fwd(x, res, gamma, beta, None, 1e-5, p, 7)
        dx, dres, dg, db, dxs = L.add_ln_bwd(dy, x, res, gamma, None, mean, rstd, p, 7, want_dx_colsum=True)
        assert rel_err(dxs, dx.double().sum(0)) <= (1e-5 if dtype == torch.float32 else 5e-3)


def test_cast2d_multi_and_derived_weights(L):
    """Stacked / transposed bf16 working copies and their refresh after an in-place update."""
    from specialk_amd import ops
    g = torch.Generator().manual_seed(21)
    Ws = [torch.nn.Parameter(torch.randn(r, 40, generator=g).to(DEV)) for r in (24, 8, 33)]
    conv = torch.nn.Parameter(torch.randn(50, 40, 1, generator=g).to(DEV))  # Conv1d(k=1) weight
    bs = [torch.nn.Parameter(torch.randn(r, generator=g).to(DEV)) for r in (24, 8, 33)]

    def check():
        cat = ops.derived_weight("cat", tuple(Ws))
        catT = ops.derived_weight("catT", tuple(Ws))
        cT = ops.derived_weight("catT", (conv,))
        bias = ops.derived_weight("bias", tuple(bs))
        ref = torch.cat([w.detach() for w in Ws], 0)
        assert cat.dtype == torch.bfloat16 and torch.equal(cat, ref.to(torch.bfloat16))
        assert catT.shape == (40, 65) and torch.equal(catT, ref.t().to(torch.bfloat16))
        assert torch.equal(cT, conv.detach().view(50, 40).t().to(torch.bfloat16))
        assert bias.dtype == torch.float32 and torch.equal(bias, torch.cat([b.detach() for b in bs]))
        return cat

    first = check()
    with torch.no_grad():
        Ws[1].add_(1.0)            # torch-side change: version counter moves, copy is rebuilt on access
    assert check().data_ptr() == first.data_ptr()
    Ws[2].data.view(-1)[:5] = 7.0   # .data write: invisible to the version counter, like the Adam kernel
    ops.refresh_derived()
    check()


@pytest.mark.parametrize("dtype,V", [(torch.float32, 23), (torch.bfloat16, 4096), (torch.bfloat16, 35000)])
def test_xent_device_gradient_scale(L, dtype, V):
    """A one-element device tensor multiplies the gradient (not the loss) in both loss kernels — the same bits
    as the host-side scale of the same value."""
    g = torch.Generator().manual_seed(V + 1)
    T = 33
    logits = rnd((T, V), dtype, g, 3.0).to(DEV)
    gold = torch.randint(1, V, (T,), generator=g)
    gold[::5] = 0
    gold = gold.to(DEV)
    scale = torch.tensor([0.6875], dtype=torch.float32, device=DEV)
    outs = []
    for kw in ({"grad_scale": 0.6875}, {"grad_scale_dev": scale}):
        dl = torch.empty_like(logits)
        stats = torch.zeros(3, dtype=torch.float64, device=DEV)
        L.xent_fwd_bwd(logits, gold, dl, stats, 1.0, 0.1, 0, **kw)
        outs.append((dl, stats))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dl1 = torch.empty_like(logits)
    L.xent_fwd_bwd(logits, gold, dl1, torch.zeros(3, dtype=torch.float64, device=DEV), 1.0, 0.1, 0)
    assert rel_err(outs[1][0], dl1.float() * 0.6875) <= (1e-6 if dtype == torch.float32 else 8e-3)


def test_empty_inputs_and_argument_errors(L):
    """Zero-sized work is accepted and touches nothing; malformed arguments come back as SK_ERR_ARG with a
    message (raised as RuntimeError by the binding) instead of launching."""
    lib = L.load()
    z = torch.zeros(8, 8, dtype=torch.bfloat16, device=DEV)
    out = torch.full((8, 8), 7.0, dtype=torch.bfloat16, device=DEV)
    L.gemm(L.GEMM_NT, z, z, 0, 8, 8, 8, 8, out=out, ldc=8)
    L.gemm(L.GEMM_NT, z, z, 8, 0, 8, 8, 8, out=out, ldc=8)
    assert (out == 7).all()
    stats = torch.zeros(3, dtype=torch.float64, device=DEV)
    L.xent_fwd_bwd(torch.empty(0, 16, dtype=torch.float32, device=DEV), torch.empty(0, dtype=torch.int64, device=DEV),
                   None, stats, 1.0, 0.1, 0)
    assert stats.abs().sum().item() == 0
    assert L.softmax_rows(torch.empty(0, 16, dtype=torch.float32, device=DEV)).shape == (0, 16)
    L.gemm_grouped_tn([])
    emb = L.embed_fwd(torch.empty(0, dtype=torch.int64, device=DEV), torch.empty(0, dtype=torch.int64, device=DEV),
                      torch.zeros(5, 8, device=DEV), torch.zeros(5, 8, device=DEV), torch.float32)
    assert emb.shape == (0, 8)
    # all-PAD rows: every loss row ignored, zero gradient
    logits = torch.randn(4, 16, device=DEV)
    dl = torch.full_like(logits, 3.0)
    L.xent_fwd_bwd(logits, torch.zeros(4, dtype=torch.int64, device=DEV), dl, stats, 1.0, 0.1, 0)
    assert stats.abs().sum().item() == 0 and dl.abs().sum().item() == 0
    # errors
    with pytest.raises(RuntimeError, match="leading dimension"):
        L.gemm(L.GEMM_NT, z, z, 8, 8, 8, 4, 8, out=out, ldc=8)
    with pytest.raises(RuntimeError, match="bad V"):
        L.xent_fwd_bwd(torch.zeros(2, 1, device=DEV), torch.zeros(2, dtype=torch.int64, device=DEV), None, stats, 1.0, 0.0, 0)
    with pytest.raises(RuntimeError, match="bad shape"):
        lib_rc = lib.skhip_softmax_rows(z.data_ptr(), z.data_ptr(), 4, 8, 4, 8, 1, None)
        L._check(lib_rc, "skhip_softmax_rows")
    with pytest.raises(RuntimeError, match="bad dtype"):
        L._check(lib.skhip_softmax_rows(z.data_ptr(), z.data_ptr(), 4, 8, 8, 8, 9, None), "skhip_softmax_rows")
    with pytest.raises(RuntimeError):  # CPU tensors never reach a kernel
        L.softmax_rows(torch.zeros(2, 4))


def test_device_collate_matches_host_collate(L):
    from specialk_amd.core.dataset import collate_fn, device_collate_fn, paired_collate_fn, paired_device_collate_fn
    rng = np.random.default_rng(2)
    insts = [[2] + rng.integers(0, 30, size=int(n)).tolist() + [3] for n in rng.integers(0, 17, size=23)]  # ids may be PAD
    seq, pos = collate_fn(insts)
    dseq, dpos = device_collate_fn(insts)
    assert dseq.is_cuda and torch.equal(dseq.cpu(), seq) and torch.equal(dpos.cpu(), pos)
    pairs = list(zip(insts, insts[::-1]))
    for a, b in zip(paired_collate_fn(pairs), paired_device_collate_fn(pairs)):
        assert torch.equal(a, b.cpu())
    one = device_collate_fn([[2, 3]])
    assert one[0].tolist() == [[2, 3]] and one[1].tolist() == [[1, 2]]


def test_full_size_generator_and_loss_kernels(L):
    """The two largest launches of the training step at their benchmark size (25 639 x 35 000, K = 512, bf16):
    sampled rows of the generator product against an fp32 matmul of the same rows, and properties of the fused
    loss kernel that hold for any input — every gradient row sums to zero (softmax and the smoothed target
    both sum to one), pad rows are zero, the counts are exact, in place equals out of place."""
    g = torch.Generator(device=DEV).manual_seed(3)
    T, V, d = 25639, 35000, 512
    x = (torch.randn(T, d, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    W = (torch.randn(V, d, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    logits = L.gemm(L.GEMM_NT, x, W, T, V, d, d, d)
    rows = torch.randint(0, T, (48,), device=DEV, generator=g)
    rows[0], rows[1] = 0, T - 1
    ref = x[rows].float() @ W.float().t()
    assert rel_err(logits[rows].float(), ref) <= BF16_RTOL / 4
    gold = torch.randint(1, V, (T,), device=DEV, generator=g)
    gold[::9] = 0
    stats = torch.zeros(3, dtype=torch.float64, device=DEV)
    dl = torch.empty_like(logits)
    L.xent_fwd_bwd(logits, gold, dl, stats, 1.0, 0.1, 0)
    n_real = int((gold != 0).sum())
    assert int(stats[2].item()) == n_real and 0 <= int(stats[1].item()) <= n_real
    assert stats[0].item() > 0 and abs(stats[0].item() / n_real - np.log(V)) < 1.0  # near-uniform predictions
    row_sums = dl.float().sum(1)
    assert row_sums.abs().max().item() <= 2e-3  # 35 000 bf16 roundings of values around 3e-5
    assert dl[gold == 0].abs().max().item() == 0
    lse = torch.logsumexp(logits[rows].float(), 1, keepdim=True)
    q = torch.full((48, V), 0.1 / (V - 1), device=DEV)
    q[torch.arange(48, device=DEV), gold[rows]] = 0.9
    ref_g = (torch.exp(logits[rows].float() - lse) - q) * (gold[rows] != 0)[:, None]
    assert rel_err(dl[rows].float(), ref_g) <= BF16_RTOL
    stats2 = torch.zeros(3, dtype=torch.float64, device=DEV)
    L.xent_fwd_bwd(logits, gold, logits, stats2, 1.0, 0.1, 0)
    assert torch.equal(logits, dl) and stats2[1].item() == stats[1].item() and stats2[2].item() == stats[2].item()


@pytest.mark.parametrize("causal", [False, True])
def test_full_size_attention_sampled_sentences(L, causal):
    """Fused attention forward / backward at the benchmark shape (640 sentences x 50 positions x 8 heads x 64,
    bf16, lengths 28..50): sentences do not interact, so a sample of them is checked against the fp64 oracle
    run on those sentences alone."""
    g = torch.Generator().manual_seed(17)
    B, Lq, H, dk = 640, 50, 8, 64
    lens = torch.randint(28, 51, (B,), generator=g)
    lens[0], lens[1] = 50, 28
    keypad = torch.arange(Lq)[None, :] >= lens[:, None]
    q, k, v, do = (rnd((B, Lq, H, dk), torch.bfloat16, g) for _ in range(4))
    o, _, _, bwd = run_attn(L, q, k, v, keypad, causal, torch.bfloat16)
    pick = torch.tensor([0, 1, 2, 317, 638, 639])
    ro, _, rbwd = oracle_attn(q[pick].float(), k[pick].float(), v[pick].float(), keypad[pick], causal)
    live = (~keypad[pick])[:, :, None, None]  # query rows past a sentence's end are padding: not compared
    assert rel_err(o.cpu()[pick].float() * live, ro * live) <= BF16_RTOL
    do_live = do.clone()  # padding query rows get no upstream gradient, as after the pad-row zeroing of the layers
    do_live[keypad] = 0
    dq, dk_, dv_ = bwd(do_live)
    rdq, rdk, rdv = rbwd(do_live[pick].float())
    assert rel_err(dq.cpu()[pick].float() * live, rdq * live) <= BF16_RTOL
    assert rel_err(dk_.cpu()[pick].float(), rdk) <= BF16_RTOL
    assert rel_err(dv_.cpu()[pick].float(), rdv) <= BF16_RTOL


def test_full_size_weight_gradients_layernorm_and_adam(L):
    """The remaining large launches of the training step at their benchmark size (25 639 token rows, d = 512,
    F = 2048): one grouped launch with the weight-gradient products of a decoder layer (contraction over all
    rows, written and accumulated), the fused residual + LayerNorm forward / backward over all rows, and one
    Adam launch over a 17.9 M-element and a 1 M-element tensor — each against a plain torch computation in
    fp32 / fp64 on the same operands."""
    g = torch.Generator(device=DEV).manual_seed(23)
    T, d, F = 25639, 512, 2048
    rn = lambda *shape, s=1.0: (torch.randn(shape, device=DEV, generator=g) * s).to(torch.bfloat16)
    # grouped weight gradients
    x, dqkv, h, dh, dy = rn(T, d), rn(T, 3 * d), rn(T, F), rn(T, F), rn(T, d)
    probs, refs = [], []
    for A, B_, acc in ((dqkv[:, :d], x, False), (dqkv[:, d:2 * d], x, True), (dqkv[:, 2 * d:], x, False), (dy, x, False),
                       (dh, x, True), (dy, h, False)):
        M, N = A.shape[1], B_.shape[1]
        C = torch.randn((M, N), device=DEV, generator=g) if acc else torch.full((M, N), float("nan"), device=DEV)
        refs.append((C, (C.double() if acc else 0) + A.double().t() @ B_.double()))
        cs = torch.full((M,), float("nan"), device=DEV)  # the bias gradient that goes with the weight gradient
        refs.append((cs, A.double().sum(0)))
        probs.append((A, B_, C, M, N, T, A.stride(0), B_.stride(0), N, acc, cs, False))
    L.gemm_grouped_tn(probs)
    for i, (C, ref) in enumerate(refs):
        assert rel_err(C, ref) <= 2e-5, i
    # residual + LayerNorm
    gamma = 1 + 0.1 * torch.randn(d, device=DEV, generator=g)
    beta = 0.1 * torch.randn(d, device=DEV, generator=g)
    pad = (torch.rand(T, device=DEV, generator=g) < 0.1)
    res = rn(T, d)
    y, mean, rstd = L.add_ln_fwd(x, res, gamma, beta, pad.to(torch.uint8), 1e-5, 0.0, 0)
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr + rr, (d,), gr, br, 1e-5) * (~pad)[:, None]
    assert rel_err(y, ref) <= BF16_RTOL and (y[pad] == 0).all()
    ref.backward(dy.double())
    dx, dres, dg, db = L.add_ln_bwd(dy, x, res, gamma, pad.to(torch.uint8), mean, rstd, 0.0, 0)
    assert rel_err(dres, rr.grad) <= BF16_RTOL
    assert rel_err(dg, gr.grad) <= 5e-3 and rel_err(db, br.grad) <= 5e-3
    # Adam, two steps
    ps = [torch.randn(35000, d, device=DEV, generator=g), torch.randn(F, d, device=DEV, generator=g)]
    gs = [torch.randn(p.shape, device=DEV, generator=g) * 1e-3 for p in ps]
    ref_p = [p.double().clone() for p in ps]
    ref_m = [torch.zeros_like(p) for p in ref_p]
    ref_v = [torch.zeros_like(p) for p in ref_p]
    ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    copies = [torch.empty(p.shape, dtype=torch.bfloat16, device=DEV) for p in ps]
    tab = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=DEV)
    tables = {"p": tab(ps), "g": tab(gs), "m": tab(ms), "v": tab(vs), "b": tab(copies),
              "n": torch.tensor([p.numel() for p in ps], dtype=torch.int64, device=DEV)}
    b1, b2, eps, lr = 0.9, 0.98, 1e-9, 3e-4
    for step in (1, 2):
        L.adam_multi(tables, 2, max(p.numel() for p in ps), lr, b1, b2, eps, step)
        for p, gg, m_, v_ in zip(ref_p, gs, ref_m, ref_v):
            m_.mul_(b1).add_(gg.double(), alpha=1 - b1)
            v_.mul_(b2).addcmul_(gg.double(), gg.double(), value=1 - b2)
            p.addcdiv_(m_ / (1 - b1 ** step), (v_ / (1 - b2 ** step)).sqrt() + eps, value=-lr)
    for p, rp, c in zip(ps, ref_p, copies):
        assert rel_err(p, rp) <= 1e-6
        assert torch.equal(c, p.to(torch.bfloat16))


def test_reserved_cus_change_scheduling_not_results(L):
    """skhip_set_reserved_cus: CUs left out of the GEMMs' round planning (held by communication kernels meanwhile)
    change tile heights, K splits and grid sizes — not the values: LayerNorm and the loss kernel bit for bit against
    the unreserved run; the products, whose number of K splits follows the CU count when their tiles do not fill the
    chip, to fp32 summation-order noise (one bf16 rounding step for bf16 outputs); argument errors."""
    g = torch.Generator().manual_seed(21)
    M = 6000
    x, w1, w2 = rnd((M, 512), torch.bfloat16, g).to(DEV), rnd((512, 512), torch.bfloat16, g).to(DEV), rnd((2048, 512), torch.bfloat16, g).to(DEV)
    dy = rnd((M, 2048), torch.bfloat16, g).to(DEV)
    res = rnd((M, 512), torch.bfloat16, g).to(DEV)
    gamma, beta = torch.ones(512, device=DEV), torch.zeros(512, device=DEV)
    logits = rnd((700, 4096), torch.bfloat16, g).to(DEV)
    gold = torch.randint(1, 4096, (700,), generator=g).to(DEV)

    def run():
        out = [L.gemm(L.GEMM_NT, x, w1, M, 512, 512, 512, 512), L.gemm(L.GEMM_NT, x, w2, M, 2048, 512, 512, 512),
               L.gemm(L.GEMM_TN, dy, x, 2048, 512, M, 2048, 512, out_dtype=torch.float32)]
        gw = torch.zeros((2048, 512), dtype=torch.float32, device=DEV)
        gb = torch.zeros(2048, dtype=torch.float32, device=DEV)
        L.gemm_grouped_tn([(dy, x, gw, 2048, 512, M, 2048, 512, 512, False, gb, False)])
        out += [gw, gb, L.add_ln_fwd(x, res, gamma, beta, None, 1e-5, 0.1, 7)[0]]
        dl, stats = torch.empty_like(logits), torch.zeros(3, dtype=torch.float64, device=DEV)
        L.xent_fwd_bwd(logits, gold, dl, stats, 1.0, 0.1, 0)
        return out + [dl, stats]

    try:
        ref = run()
        for n in (8, 40, 200):
            L.set_reserved_cus(n)
            for i, (a, b) in enumerate(zip(run(), ref)):
                if i < 5:   # products: the number of K splits follows the CU count (bf16 outputs: one rounding step)
                    assert rel_err(a, b.double()) <= (5e-3 if a.dtype == torch.bfloat16 else 1e-5), (n, i)
                else:
                    assert torch.equal(a, b), (n, i)
        for bad in (-1, 10 ** 6):
            with pytest.raises(RuntimeError):
                L.set_reserved_cus(bad)
        # the row-panel kernels take one CU per 104 rows for a whole launch: with CUs held they are only chosen when
        # their workgroups still fill the rounds they take
        from specialk_amd import ops
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        rows = torch.empty((104 * (cus - 9), 512), dtype=torch.bfloat16, device=DEV)   # a full round less nine CUs
        L.set_reserved_cus(0)
        assert ops._panel_rows_ok(rows, rows.shape[0])
        L.set_reserved_cus(16)
        assert L.reserved_cus == 16 and not ops._panel_rows_ok(rows, rows.shape[0])
    finally:
        L.set_reserved_cus(0)


@pytest.mark.parametrize("R,V,layout", [(640, 35000, 1), (640, 4999, 0), (35, 1003, 1), (35, 1003, 0), (5, 50, 1), (300, 144, 1),
                                        (129, 1441, 0)])
def test_generator_rowstats_equals_tiled_product(L, R, V, layout):
    """skhip_generator_rowstats (a workgroup per 144 vocabulary entries, every row) against skhip_gemm with
    SK_EPI_ROWSTATS: the fp32 products bit for bit (same MFMA, same order over K), the statistics of its 48-column
    chunks against the products themselves, and the same beam choices through skhip_beam_topk_stats with either
    chunk width; activations row-major and in the fragment-major order (fragment_major_rows and back), vocabularies
    that end inside a chunk, inside a tile and (V = 144) with a slab."""
    g = torch.Generator().manual_seed(7 * R + V)
    d, bm = 512, 5
    x = rnd((R, d), torch.bfloat16, g).to(DEV)
    W = (rnd((V, d), torch.bfloat16, g) * 0.2).to(DEV)
    cw = L.generator_chunk_cols()
    assert cw == 48
    ref = torch.empty((R, V), dtype=torch.float32, device=DEV)
    st_ref = torch.empty((R, (V + 63) // 64, 2), dtype=torch.float32, device=DEV)
    L.gemm(L.GEMM_NT, x, W, R, V, d, d, d, out=ref, ldc=V, aux=st_ref, epilogue=L.EPI_ROWSTATS, split_k=False)
    got = torch.full((R, V), float("nan"), dtype=torch.float32, device=DEV)
    nch = (V + cw - 1) // cw
    st = torch.full((R, nch, 2), float("nan"), dtype=torch.float32, device=DEV)
    xs = L.fragment_major_rows(x) if layout else x
    if layout:
        assert torch.equal(L.plain_rows(xs, R), x)
    L.generator_rowstats(xs, W, got, st, layout)
    assert torch.equal(got, ref)
    pad = torch.full((R, nch * cw), float("-inf"), dtype=torch.float32, device=DEV)
    pad[:, :V] = got
    ch = pad.view(R, nch, cw)
    mx = ch.max(2).values
    assert torch.equal(st[:, :, 0], mx)
    assert rel_err(st[:, :, 1], torch.exp(ch - mx.unsqueeze(2)).sum(2)) <= 1e-5
    if R % bm == 0 and V >= bm:
        n = R // bm
        scores = -torch.rand(n, bm, generator=g).to(DEV)
        for first in (True, False):
            a = tuple(torch.empty((n, bm), dtype=t, device=DEV) for t in (torch.float32, torch.int32, torch.int32))
            b = tuple(torch.empty((n, bm), dtype=t, device=DEV) for t in (torch.float32, torch.int32, torch.int32))
            L.beam_topk_stats(ref, st_ref, scores, n, bm, first, a, chunk_cols=64)
            L.beam_topk_stats(got, st, scores, n, bm, first, b, chunk_cols=cw)
            assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
            assert (a[0] - b[0]).abs().max().item() <= 2e-5


def test_generator_rowstats_rejects_other_widths(L):
    x = torch.zeros((16, 256), dtype=torch.bfloat16, device=DEV)
    W = torch.zeros((100, 256), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="K must be 512"):
        L.generator_rowstats(x, W, torch.empty((16, 100), device=DEV), torch.empty((16, 3, 2), device=DEV))
