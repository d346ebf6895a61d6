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


@pytest.mark.parametrize("w8", ["0", "1"])
@pytest.mark.parametrize("gk,near", [(4, 2), (1, 1), (8, 3), (2, 1), (3, 5)])
def test_cholesky_task_list_is_a_topological_order(gk, near, w8, monkeypatch):
    """The static task list of the one-launch Cholesky (alabi/core.py:1158 -> gp.compute): replayed in order on the host,
    every task finds its inputs produced by EARLIER tasks (a workgroup only ever waits for lower-numbered tasks: no deadlock),
    every tile receives every block column exactly once and in order, and every panel tile is solved exactly once."""
    from alabi_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.alabi_debug_chol_tasks.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    monkeypatch.setenv("ALABI_CHOL_GK", str(gk))
    monkeypatch.setenv("ALABI_CHOL_NEAR", str(near))
    monkeypatch.setenv("ALABI_CHOL_W8", w8)                  # the eight-wave kernel's list holds UPDATE2 tasks (two tiles per grouped update)
    for nb in (3, 5, 16, 33, 79):
        n = lib.alabi_debug_chol_tasks(nb, None, 0)
        buf = (ctypes.c_int * (4 * n))()
        assert lib.alabi_debug_chol_tasks(nb, buf, n) == n
        ver = [[0] * nb for _ in range(nb)]                  # block columns applied to tile (i, j)
        final = [[False] * nb for _ in range(nb)]            # tile (i, j) holds its block of L
        for q in range(n):
            ty, i, j, k = buf[4 * q] & 255, buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]
            cnt = buf[4 * q] >> 8
            if ty == 0:                                      # CHAIN(k): solve (k, k-1), update + factorise (k, k)
                assert i == j == k and not final[k][k]
                if k > 0:
                    assert final[k - 1][k - 1] and ver[k][k - 1] == k - 1 and not final[k][k - 1]
                    assert ver[k][k] == k - 1
                    final[k][k - 1] = True
                    ver[k][k] = k
                final[k][k] = True
            elif ty == 1:                                    # TRSM(i, k)
                assert j == k and i >= k + 2 and final[k][k] and ver[i][k] == k and not final[i][k]
                final[i][k] = True
            else:                                            # UPDATE(i, j, k .. k + cnt - 1); UPDATE2 = the same for tile rows i and i + 1
                assert ty in (2, 4, 5) and cnt >= 1 and i >= j > k + cnt - 1      # UPDATE4 = the 2 x 2 block of tiles from (i, j)
                assert ty == 2 or (cnt >= 2 and i + 1 < nb)
                assert ty != 5 or (i >= j + 1 and j + 1 < nb)
                for c in ((j,) if ty != 5 else (j, j + 1)):
                    for r in ((i,) if ty == 2 else (i, i + 1)):
                        assert r >= c and not (r == c == k + cnt)   # lower triangle; the last column of a diagonal tile belongs to CHAIN
                        assert ver[r][c] == k and final[r][k + cnt - 1] and final[c][k + cnt - 1]
                        ver[r][c] = k + cnt
        for i in range(nb):
            for j in range(i + 1):
                assert final[i][j] and ver[i][j] == j, (nb, i, j)
