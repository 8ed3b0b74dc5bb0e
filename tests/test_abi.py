"""CPU: the C-ABI library builds, loads, and exports exactly the symbols ``include/sqgr.h`` declares."""

from __future__ import annotations

import ctypes
import os
import re

import pytest

from squidpy_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared() -> set[str]:
    with open(os.path.join(ROOT, "include", "sqgr.h")) as fh:
        src = fh.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(sqgr_[a-z0-9_]+)\s*\(", src))


@pytest.fixture(scope="module")
def lib_path():
    return _build.build(verbose=False)


def test_header_symbols_are_exported_and_bound(lib_path):
    declared = _declared()
    assert declared, "no declarations parsed from include/sqgr.h"
    lib = ctypes.CDLL(lib_path)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in sqgr.h but not exported by libsqgr.so"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_library_loads_and_reports_abi(lib_path):
    lib = _lib.load_library()
    assert lib.sqgr_abi_version() == 7
    assert isinstance(lib.sqgr_last_error(), bytes)


def test_no_silent_cpu_fallback_without_device(lib_path):
    """Without a HIP device context creation must fail loudly (there is no CPU path in the product)."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.SqgrError):
        _lib.Context(0)


def test_product_does_not_import_oracle():
    """The product package must never import, call or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "squidpy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dirpath, f)) as fh:
                    src = fh.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_every_entry_point_is_named_in_the_integration_guide():
    """INTEGRATION.md shows the reference-side binding: every function of the boundary appears there (next to the reference call
    site it replaces, or in the list of lifecycle / introspection / parity-hook entry points)."""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as fh:
        guide = fh.read()
    missing = sorted(name for name in _declared() if name not in guide)
    assert not missing, missing
