"""The allocation-free column-sum entry point (skhip_colsum_noalloc) against skhip_colsum, through
the C ABI: bit-identical results on two calls in a row, the second with a smaller input, so that
stale partial sums of the first call in the shared built-in scratch array would show."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

SCRATCH_BYTES = 2 << 20  # the built-in array's size: the same workspace gives skhip_colsum the same chunking


def _both(lib, X, M, N, ldx, accumulate, init):
    from specialk_amd import _lib as L
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(SCRATCH_BYTES, dtype=torch.uint8, device="cuda")
    ref, got = init.clone(), init.clone()
    code = L.dcode(X.dtype)
    rc = lib.skhip_colsum(X.data_ptr(), ref.data_ptr(), M, N, ldx, accumulate, code, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, lib.skhip_last_error()
    rc = lib.skhip_colsum_noalloc(X.data_ptr(), got.data_ptr(), M, N, ldx, accumulate, code, stream)
    assert rc == 0, lib.skhip_last_error()
    torch.cuda.synchronize()
    return ref, got


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_colsum_noalloc_matches_colsum_twice(dtype):
    from specialk_amd import _lib as L
    lib = L.load()
    assert lib.skhip_init() == 0, lib.skhip_last_error()
    g = torch.Generator().manual_seed(11)
    # first call: many row chunks, wide; second call: fewer rows and columns (a sub-view with a row stride)
    for (M, N, ld, acc) in [(20000, 1536, 1536, 0), (777, 136, 200, 1), (64, 8, 8, 0)]:
        X = torch.randn((M, ld), generator=g).to(dtype).cuda()
        init = torch.randn(N, generator=g).cuda()
        ref, got = _both(lib, X, M, N, ld, acc, init)
        assert torch.equal(ref, got), (M, N, ld, acc, (ref - got).abs().max().item())
        exact = X[:, :N].double().sum(0) + (init.double() if acc else 0)
        assert ((got.double() - exact).abs().max() <= 1e-3 * exact.abs().max().clamp_min(1.0)).item()


def test_colsum_noalloc_rejects_oversize():
    from specialk_amd import _lib as L
    lib = L.load()
    X = torch.zeros((2, 8), device="cuda")
    out = torch.zeros(8, device="cuda")
    rc = lib.skhip_colsum_noalloc(X.data_ptr(), out.data_ptr(), 2, 65536 + 8, 65536 + 8, 0, 0, None)
    assert rc == -1 and b"outside" in lib.skhip_last_error()


def test_colsum_noalloc_allocates_nothing():
    from specialk_amd import _lib as L
    lib = L.load()
    X = torch.randn((5000, 512), device="cuda").bfloat16()
    out = torch.empty(512, device="cuda")
    L.colsum_noalloc(X, 5000, 512, 512, out)  # first call on the device looks the array's address up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    for _ in range(3):
        L.colsum_noalloc(X, 5000, 512, 512, out)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
    assert torch.cuda.mem_get_info()[0] == free0
