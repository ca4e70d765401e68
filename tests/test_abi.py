"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, and the header and the ctypes binding agree on the symbol list.  No compute calls."""
import ctypes
import os
import re

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "specialk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skhip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from specialk_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `make -C specialk_amd/csrc` first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/specialk_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == syms


def test_version_and_error_strings():
    from specialk_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.skhip_version()
    assert isinstance(lib.skhip_last_error(), bytes)


def test_no_cpu_fallback():
    """CPU tensors are rejected loudly instead of being computed some other way."""
    import pytest
    import torch
    from specialk_amd import _lib
    a = torch.zeros(4, 4)
    with pytest.raises(RuntimeError):
        _lib.gemm(_lib.GEMM_NT, a, a, 4, 4, 4, 4, 4, split_k=False)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "specialk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
