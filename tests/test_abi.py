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


def _replay_cholesky_tasks(tasks, nb):
    """Replay one matrix's task list (quadruples (type, i, j, k), the type's bits 8..15 = block columns per UPDATE) on the host:
    every task must find its inputs produced by EARLIER tasks; every tile receives every block column exactly once, in order."""
    ver = [[0] * nb for _ in range(nb)]                  # block columns applied to tile (i, j)
    final = [[False] * nb for _ in range(nb)]            # tile (i, j) holds its block of L
    for t, i, j, k in tasks:
        ty, cnt = t & 255, (t >> 8) & 255
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


def _single_list(lib, nb):
    n = lib.alabi_debug_chol_tasks(nb, None, 0)
    buf = (ctypes.c_int * (4 * n))()
    assert lib.alabi_debug_chol_tasks(nb, buf, n) == n
    return [(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]) for q in range(n)]


def _batch_matrix_list(lib, nb):
    n = lib.alabi_debug_chol_batch_matrix_tasks(nb, None, 0)
    buf = (ctypes.c_int * (4 * n))()
    assert lib.alabi_debug_chol_batch_matrix_tasks(nb, buf, n) == n
    return [(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]) for q in range(n)]


def _debug_lib():
    from alabi_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.alabi_debug_chol_tasks.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.alabi_debug_chol_batch_matrix_tasks.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.alabi_debug_chol_batch_tasks.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    return lib


@pytest.mark.parametrize("w8", ["0", "1"])
@pytest.mark.parametrize("gk,near", [(4, 2), (1, 1), (8, 3), (2, 1), (3, 5)])
def test_cholesky_task_list_is_a_topological_order(gk, near, w8, monkeypatch):
    """The static task list of the one-launch Cholesky (alabi/core.py:1158 -> gp.compute): replayed in order on the host,
    every task finds its inputs produced by EARLIER tasks (a workgroup only ever waits for lower-numbered tasks: no deadlock),
    every tile receives every block column exactly once and in order, and every panel tile is solved exactly once."""
    lib = _debug_lib()
    monkeypatch.setenv("ALABI_CHOL_GK", str(gk))
    monkeypatch.setenv("ALABI_CHOL_NEAR", str(near))
    monkeypatch.setenv("ALABI_CHOL_W8", w8)                  # the eight-wave kernel's list holds UPDATE2 tasks (two tiles per grouped update)
    for nb in (3, 5, 16, 33, 79):
        _replay_cholesky_tasks(_single_list(lib, nb), nb)


@pytest.mark.parametrize("nb,update4", [(100, None), (157, None), (256, None), (21, "1"), (40, "1"), (120, "0")])
def test_cholesky_task_list_default_switches_and_update4(nb, update4, monkeypatch):
    """The lists the library actually selects (default group size / near band / eight waves) at the sizes where the far updates take
    2 x 2 tiles per task (UPDATE4, default from 100 block columns = N >= 6400) up to the queue's limit of 256 block columns, and
    UPDATE4 forced on at small sizes / off at a large one: the same replay."""
    lib = _debug_lib()
    for k in ("ALABI_CHOL_GK", "ALABI_CHOL_NEAR", "ALABI_CHOL_W8", "ALABI_CHOL_UPDATE2"):
        monkeypatch.delenv(k, raising=False)
    if update4 is None:
        monkeypatch.delenv("ALABI_CHOL_UPDATE4", raising=False)
    else:
        monkeypatch.setenv("ALABI_CHOL_UPDATE4", update4)
    tasks = _single_list(lib, nb)
    has4 = any((t[0] & 255) == 5 for t in tasks)
    assert has4 == (update4 == "1" or (update4 is None and nb >= 100))
    _replay_cholesky_tasks(tasks, nb)


@pytest.mark.parametrize("gk", [1, 2, 3, 4, 8, 12, 16, 255])
@pytest.mark.parametrize("update4", ["0", "1"])
def test_batched_cholesky_matrix_list_is_a_topological_order(gk, update4, monkeypatch):
    """The list of ONE matrix inside a batch (chol_build_tasks_batch: full groups while the chain is far, one catch-up task per
    tile, left-looking for gk >= nb): the same host replay -- inputs produced by earlier tasks, every block column exactly once and
    in order on every tile."""
    lib = _debug_lib()
    monkeypatch.setenv("ALABI_BATCH_GK", str(gk))
    monkeypatch.setenv("ALABI_CHOL_UPDATE4", update4)
    monkeypatch.delenv("ALABI_BATCH_LEFT", raising=False)
    for nb in (1, 2, 3, 4, 5, 7, 12, 13, 16, 25, 26, 40, 79):
        tasks = _batch_matrix_list(lib, nb)
        _replay_cholesky_tasks(tasks, nb)
        if gk >= nb:                                     # left-looking: every tile that takes updates is written by ONE task
            tiles = sum({2: 1, 4: 2, 5: 4}[t[0] & 255] for t in tasks if (t[0] & 255) >= 2)
            assert tiles == (nb - 1) * (nb - 2) // 2 + max(nb - 2, 0)     # (i, j) with i > j >= 1, and the diagonal tiles from (2, 2) on


@pytest.mark.parametrize("phases", ["4", "3", "0"])
@pytest.mark.parametrize("left", ["1", "0"])
@pytest.mark.parametrize("nlists,window", [(1, 0), (1, 3), (3, 2), (8, 3), (8, 0), (8, 100)])
def test_batched_cholesky_queue_keeps_every_matrix_in_order(nlists, window, left, phases, monkeypatch):
    """The interleaved queue of the batched factorisation (alabi/gp_utils.py:511-700: candidates x folds): every matrix lives in
    exactly one list, its tasks appear there in the order of its own single-matrix list (so each list is a topological order and a
    workgroup only waits for tasks in front of the one it drew), nothing is lost or duplicated -- for mixed sizes, any number of
    lists and any stagger."""
    lib = _debug_lib()
    for k in ("ALABI_CHOL_GK", "ALABI_CHOL_NEAR", "ALABI_CHOL_W8", "ALABI_CHOL_UPDATE2", "ALABI_CHOL_UPDATE4", "ALABI_BATCH_GK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("ALABI_BATCH_LEFT", left)
    monkeypatch.setenv("ALABI_BATCH_PHASES", phases)                 # slot order: phase by phase over the matrices / matrix by matrix
    nbs = [25, 25, 26, 3, 25, 16, 40, 25, 1, 5, 25, 2, 33, 25, 25, 7, 25]
    B = len(nbs)
    arr = (ctypes.c_int * B)(*nbs)
    n = lib.alabi_debug_chol_batch_tasks(B, arr, nlists, window, None, 0, None)
    buf = (ctypes.c_int * (4 * n))()
    lo = (ctypes.c_int * (nlists + 1))()
    assert lib.alabi_debug_chol_batch_tasks(B, arr, nlists, window, buf, n, lo) == n
    assert lo[0] == 0 and lo[nlists] == n and all(lo[q] <= lo[q + 1] for q in range(nlists))
    per_matrix = {b: [] for b in range(B)}
    home = {}
    for q in range(nlists):
        for x in range(lo[q], lo[q + 1]):
            t = buf[4 * x]
            b = t >> 16
            assert home.setdefault(b, q) == q                 # a matrix never changes lists
            per_matrix[b].append((t & 0xFFFF, buf[4 * x + 1], buf[4 * x + 2], buf[4 * x + 3]))
    assert sorted(home) == list(range(B))
    for b in range(B):
        assert per_matrix[b] == _batch_matrix_list(lib, nbs[b])     # the matrix's own order, task for task
        if left == "0":
            assert per_matrix[b] == _single_list(lib, nbs[b])
        _replay_cholesky_tasks(per_matrix[b], nbs[b])
