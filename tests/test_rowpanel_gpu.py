"""The row-panel products (skhip_rowpanel_gemm / skhip_rowstream_gemm and their LayerNorm forms) against a float64
computation of the same lines of the reference (nn.Linear / Conv1d(k=1), SubLayers.py:46-48,62,73,80-84) on the same
bf16 inputs, and against the tiled kernels they stand in for (skhip_gemm + skhip_add_ln_fwd).  Through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF16_RTOL = 2e-2  # north-star tolerance for bf16 storage


@pytest.fixture(scope="module")
def L():
    from specialk_amd import _lib
    _lib.load()
    return _lib


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def max_rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def make(T, N, K, seed, lda=None):
    g = torch.Generator().manual_seed(seed)
    lda = K if lda is None else lda
    a_full = (torch.randn(T, lda, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    W = torch.nn.Parameter((torch.randn(N, K, generator=g) * K ** -0.5).to(DEV))
    bias = (torch.randn(N, generator=g) * 0.2).to(DEV)
    return a_full[:, :K], W, bias


def fm(W, transposed=False):
    from specialk_amd import ops
    return ops.derived_weight("fmT" if transposed else "fm", [W])


def ref_product(a, W):
    return a.double().cpu() @ W.detach().to(torch.bfloat16).double().cpu().t()


# rows: whole panels, a ragged tail, fewer rows than one panel, one row
@pytest.mark.parametrize("T", [104 * 3, 1000, 57, 1])
@pytest.mark.parametrize("N", [512, 1536])
def test_panel_plain_and_bias(L, T, N):
    a, W, bias = make(T, N, 512, T + N, lda=520)
    ref = ref_product(a, W)
    out = L.rowpanel_gemm(a, fm(W), N)
    assert max_rel(out, ref) <= BF16_RTOL
    out_b = L.rowpanel_gemm(a, fm(W), N, bias=bias)
    assert max_rel(out_b, ref + bias.double().cpu()) <= BF16_RTOL
    # the tiled kernel on the same operands: same values up to the summation order (one bf16 step)
    tiled = L.gemm(L.GEMM_NT, a, W.detach().to(torch.bfloat16), T, N, 512, a.stride(0), 512, bias=bias)
    assert max_rel(out_b, tiled) <= 1e-2


def test_panel_into_column_slice_and_transposed_weight(L):
    T, N = 700, 512
    a, W, _ = make(T, N, 512, 11)
    buf = torch.full((T, 1536), 7.0, dtype=torch.bfloat16, device=DEV)
    L.rowpanel_gemm(a, fm(W), N, out=buf[:, 512:1024], ldc=1536)
    assert max_rel(buf[:, 512:1024], ref_product(a, W)) <= BF16_RTOL
    assert bool((buf[:, :512] == 7.0).all()) and bool((buf[:, 1024:] == 7.0).all())
    # input gradient: dy W with the weight's transpose as the fragment-major operand
    Wt = torch.nn.Parameter(W.detach().t().contiguous())
    out = L.rowpanel_gemm(a, fm(Wt, transposed=True), N)
    assert max_rel(out, ref_product(a, W)) <= BF16_RTOL


@pytest.mark.parametrize("T", [104 * 2 + 9, 33])
def test_panel_relu_pattern_and_mask(L, T):
    F = 2048
    a, W1, b1 = make(T, F, 512, 3 * T)
    signs = L.rowpanel_pattern(T, F, DEV)
    h = L.rowpanel_gemm(a, fm(W1), F, bias=b1, signs=signs, epilogue=L.EPI_RELU | L.EPI_RELU_BITS_OUT)
    ref = torch.relu(ref_product(a, W1) + b1.double().cpu())
    assert max_rel(h, ref) <= BF16_RTOL
    assert float(h.min()) >= 0.0 and 0.3 < float((h > 0).float().mean()) < 0.7
    # backward of the ReLU: (dz W2) masked by the pattern (the kernels' own layout: only the masked product reads it);
    # W2 (512, F) enters transposed.  The mask must be exactly h > 0.
    _, W2, _ = make(T, 512, F, 5 * T)
    dz = (torch.randn(T, 512, generator=torch.Generator().manual_seed(T)) * 0.3).to(torch.bfloat16).to(DEV)
    dh = L.rowpanel_gemm(dz, fm(W2, transposed=True), F, signs=signs, epilogue=L.EPI_RELU_MASK_BITS)
    full = L.rowpanel_gemm(dz, fm(W2, transposed=True), F)
    assert torch.equal(dh, torch.where(h > 0, full, torch.zeros_like(full)))
    ref_dh = (dz.double().cpu() @ W2.detach().to(torch.bfloat16).double().cpu()) * (h > 0).cpu()
    assert max_rel(dh, ref_dh) <= BF16_RTOL
    # the tiled kernels with their own pattern
    signs_t = torch.zeros((T, F // 16), dtype=torch.int16, device=DEV)
    h_t = L.gemm(L.GEMM_NT, a, W1.detach().to(torch.bfloat16), T, F, 512, a.stride(0), 512, bias=b1, aux=signs_t,
                 epilogue=L.EPI_RELU | L.EPI_RELU_BITS_OUT)
    tiled = L.gemm(L.GEMM_NT, dz, W2.detach().t().contiguous().to(torch.bfloat16), T, F, 512, 512, 512, aux=signs_t,
                   epilogue=L.EPI_RELU_MASK_BITS)
    assert max_rel(h, h_t) <= 1e-2 and max_rel(dh, tiled) <= 2e-2


def test_panel_accumulate(L):
    T = 300
    a, W, _ = make(T, 512, 512, 21)
    c0 = (torch.randn(T, 512, generator=torch.Generator().manual_seed(2))).to(torch.bfloat16).to(DEV)
    c = c0.clone()
    L.rowpanel_gemm(a, fm(W), 512, out=c, ldc=512, epilogue=L.EPI_ACCUM)
    assert max_rel(c, ref_product(a, W) + c0.double().cpu()) <= BF16_RTOL


def test_stacked_transposed_weights(L):
    """fragment-major copies of several matrices: stacked by rows ([Wq; Wk; Wv], the fused projection) and their
    transposes side by side (the input-gradient operand of a stacked projection, K slices of one destination)."""
    T = 260
    g = torch.Generator().manual_seed(9)
    Ws = [torch.nn.Parameter((torch.randn(512, 512, generator=g) * 0.04).to(DEV)) for _ in range(3)]
    x = (torch.randn(T, 512, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    from specialk_amd import ops
    out = L.rowpanel_gemm(x, ops.derived_weight("fm", Ws), 1536)
    ref = x.double().cpu() @ torch.cat([w.detach() for w in Ws]).to(torch.bfloat16).double().cpu().t()
    assert max_rel(out, ref) <= BF16_RTOL
    # (512, 1536) = [Wq^T | Wk^T | Wv^T] as a fragment-major operand with N = 512, K = 1536: read back through its
    # layout formula, [N / 64][K / 32][4][64][8] with tile row i <-> column 16 (i >> 2) + 4 tile + (i & 3)
    wt = ops.derived_weight("fmT", Ws).cpu().view(8, 48, 4, 64, 8)
    cat_t = torch.cat([w.detach() for w in Ws]).to(torch.bfloat16).t().cpu()  # (512, 1536)
    for n, k in [(0, 0), (5, 40), (77, 600), (300, 1111), (511, 1535)]:
        j = n % 64
        tile, i = (j % 16) // 4, 4 * (j // 16) + (j & 3)
        lane = i + 16 * ((k % 32) // 8)
        assert wt[n // 64, k // 32, tile, lane, k % 8] == cat_t[n, k], (n, k)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_layernorm_tail_equals_separate_kernels(L, p_drop):
    """sub-layer output, LayerNorm output and the saved statistics of the fused call equal skhip_gemm followed by
    skhip_add_ln_fwd with the same dropout seed (the keep decisions are a function of seed and element index)."""
    K = 512
    T = 104 * 2 + 31
    a, W, bias = make(T, 512, K, 7 * K, lda=K + 16)
    g = torch.Generator().manual_seed(K)
    res_full = (torch.randn(T, 528, generator=g)).to(torch.bfloat16).to(DEV)
    res = res_full[:, :512]
    gamma = (1 + 0.1 * torch.randn(512, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(512, generator=g)).to(DEV)
    row_pad = (torch.rand(T, generator=g) < 0.2).to(torch.uint8).to(DEV)
    seed = 0x1234567 + K
    sub, y, mean, rstd = L.rowpanel_gemm_ln(a, fm(W), bias, res, gamma, beta, row_pad, 1e-5, p_drop, seed)
    sub_t = L.gemm(L.GEMM_NT, a, W.detach().to(torch.bfloat16), T, 512, K, a.stride(0), K, bias=bias)
    assert max_rel(sub, ref_product(a, W) + bias.double().cpu()) <= BF16_RTOL
    assert max_rel(sub, sub_t) <= 1e-2
    # LayerNorm of the fused call's own sub-layer output by the separate kernel: same rows in, same rows out
    y_t, mean_t, rstd_t = L.add_ln_fwd(sub, res.contiguous(), gamma, beta, row_pad, 1e-5, p_drop, seed)
    assert max_rel(mean, mean_t) <= 1e-5 and max_rel(rstd, rstd_t) <= 1e-5
    assert max_rel(y, y_t) <= 1e-2
    assert bool((y[row_pad.bool()] == 0).all())
    if p_drop == 0.0:  # and against float64 arithmetic
        z = res.double().cpu() + sub.double().cpu()
        ref = torch.nn.functional.layer_norm(z, (512,), gamma.double().cpu(), beta.double().cpu(), 1e-5)
        ref[row_pad.bool().cpu()] = 0
        assert max_rel(y, ref) <= BF16_RTOL


def test_full_size_rows(L):
    """25.6 k rows (one round of panels), N = 2048 and the 6144-wide key | value projection: sampled rows against
    float64."""
    T = 25639
    for N, seed in ((2048, 77), (6144, 78)):
        a, W, b = make(T, N, 512, seed)
        out = L.rowpanel_gemm(a, fm(W), N, bias=b)
        rows = torch.cat([torch.arange(0, 40), torch.arange(T - 40, T),
                          torch.randint(0, T, (200,), generator=torch.Generator().manual_seed(1))])
        assert max_rel(out[rows], ref_product(a[rows], W) + b.double().cpu()) <= BF16_RTOL
