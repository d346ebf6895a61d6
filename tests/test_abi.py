"""The C-ABI library builds, loads without a GPU, and exports every symbol include/alabi_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "alabi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alabi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared()
    assert "alabi_gp_compute" in names and "alabi_ens_run" in names and len(names) >= 25


def test_library_exports_every_declared_symbol():
    from alabi_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(handle, name), f"{name} declared in alabi_hip.h but not exported"


def test_binding_covers_header_exactly():
    from alabi_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.lib()
    assert lib.alabi_abi_version() == 1
    assert lib.alabi_status_string(1).decode() == "matrix is not positive definite"


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    from alabi_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.lib()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "alabi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
