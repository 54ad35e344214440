"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol the header declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from missm_benchmark_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "missm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(missm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/missm_hip.h but not exported"


def test_binding_covers_header():
    from missm_benchmark_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.PLAIN)
    assert bound == set(_header_symbols())


def test_argument_counts_match_header():
    from missm_benchmark_amd import _lib
    text = open(os.path.join(ROOT, "include", "missm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, args in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^)]*)\)", text)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(args), name


def test_error_channel_without_gpu(lib):
    # argument validation happens on the host before any launch, so it is testable without a GPU
    rc = lib.missm_gemm_nt(None, None, None, 0, 0, 0, 0, 0, 0, 1.0, None, None, None, None, 0, 0, 0, 0, 1, None)
    assert rc == -1
    assert b"gemm" in lib.missm_last_error()
    assert lib.missm_abi_version() == 2
    assert lib.missm_device_count() >= 0


def test_product_path_has_no_oracle_import():
    pkg = os.path.join(ROOT, "missm_benchmark_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "missm_oracle" not in src and "ref_shims" not in src, f"{f} touches the oracle"
