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


def test_default_library_cannot_skip_work():
    """The shipped libpsg.so carries no work-skipping timing switch: the PSG_DIAG bits exist only in -DPSG_DIAG_BUILD
    libraries (tools/diag_*.sh), the default build has neither the string nor the code, and says so."""
    from pointsecguard_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"PSG_DIAG" not in blob
    assert _lib.load().psg_diag_build() == 0


def test_env_switches_are_reported(monkeypatch):
    """psg_env_switches lists the registered switches that are set, with their kind; every getenv of the sources goes
    through the registry (psg::env_str / env_int), and every name read there is registered."""
    from pointsecguard_amd import _lib
    for name, _, _ in _lib.env_switches():
        monkeypatch.delenv(name)
    assert _lib.env_switches() == []
    monkeypatch.setenv("PSG_GCN_KNN", "f32")
    monkeypatch.setenv("PSG_TRACE_SYNC", "0")
    monkeypatch.setenv("PSG_DIAG", "41")          # not a switch of the default library: not listed, not read
    assert sorted(_lib.env_switches()) == [("PSG_GCN_KNN", "f32", "p"), ("PSG_TRACE_SYNC", "0", "d")]
    csrc = os.path.join(ROOT, "pointsecguard_amd", "csrc")
    api = open(os.path.join(csrc, "psg_api.hip")).read()
    registered = set(re.findall(r'\{"(PSG_[A-Z0-9_]+)", \'[pdr]\'\}', api))
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cuh", ".h")):
            src = open(os.path.join(csrc, f)).read()
            if f != "psg_api.hip":
                assert "getenv(" not in src, f
            for name in re.findall(r'env_(?:str|int)\("(PSG_[A-Z0-9_]+)"', src):
                assert name in registered, (f, name)


def test_bench_refuses_library_switches():
    """bench.py stops before any device work when a path-selecting switch is set without --allow-env-switches."""
    import subprocess
    import sys
    env = dict(os.environ, PSG_GCN_KNN="f32")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to run with non-default library switches" in (r.stderr + r.stdout)
