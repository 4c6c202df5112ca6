"""CPU tests of the C-ABI boundary: the library loads and exports exactly what include/psg.h declares
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "psg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psg_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding():
    from pointsecguard_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_library_exports_every_declared_symbol():
    from pointsecguard_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_error_strings():
    from pointsecguard_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.psg_version()
    assert isinstance(lib.psg_last_error(), bytes)


def test_no_cpu_fallback():
    """Ops must fail loudly on CPU tensors instead of silently computing elsewhere."""
    import torch
    from pointsecguard_amd import _lib, runtime
    with pytest.raises(_lib.PsgError):
        runtime.fps(torch.zeros(1, 64, 3), 4, torch.zeros(1, dtype=torch.int32))
    with pytest.raises(_lib.PsgError):
        runtime.require_cuda(torch.zeros(3), "x")


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under pointsecguard_amd/ may reference it."""
    pkg = os.path.join(ROOT, "pointsecguard_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libpsg_oracle" not in src, f
