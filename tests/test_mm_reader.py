"""CPU-only: the MatrixMarket reader of libcg_hip (include/mm_reader.h) against a pure-Python restatement of the
reference's inline reader (tests/test_vec.c:443-560 there) and against scipy.io.mmread, on the synthetic fixtures of
oracle/gen_mtx.py.  (The reference's own bcsstk*.mtx inputs are not in this repo: copying them was refused by the
environment)"""
import ctypes as C
import os

import numpy as np
import pytest

import foamio

MTX = os.path.join(foamio.GOLDEN, "mtx")
FILES = ["plate_24x18.mtx", "skewed_1500.mtx", "general_60.mtx", "pattern_40.mtx", "integer_30.mtx"]


def read_with_library(path):
    import conjugate_gradient_amd as cga
    L = cga.lib()
    dpp, ipp = C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_int))
    L.mm_read_symmetric_split.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), dpp, dpp, ipp, ipp, C.POINTER(C.c_int)]
    L.mm_free.argtypes = [C.c_void_p]
    n, U, sym = C.c_int(), C.c_int(), C.c_int()
    d, u = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    r, c = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    rc = L.mm_read_symmetric_split(path.encode(), C.byref(n), C.byref(U), C.byref(d), C.byref(u), C.byref(r), C.byref(c), C.byref(sym))
    if rc != 0:
        return rc, None
    out = {"n": n.value, "U": U.value, "symmetric": bool(sym.value),
           "diag": np.ctypeslib.as_array(d, shape=(max(n.value, 1),))[: n.value].copy(),
           "upper": np.ctypeslib.as_array(u, shape=(max(U.value, 1),))[: U.value].copy(),
           "rows": np.ctypeslib.as_array(r, shape=(max(U.value, 1),))[: U.value].copy(),
           "cols": np.ctypeslib.as_array(c, shape=(max(U.value, 1),))[: U.value].copy()}
    for p in (d, u, r, c):
        L.mm_free(p)
    return 0, out


@pytest.mark.parametrize("name", FILES)
def test_split_matches_restatement(name):
    path = os.path.join(MTX, name)
    rc, got = read_with_library(path)
    assert rc == 0
    want = foamio.mtx_split(path)
    assert (got["n"], got["U"], got["symmetric"]) == (want["n"], want["U"], want["symmetric"])
    for k in ("diag", "upper", "rows", "cols"):
        assert np.array_equal(got[k], want[k]), k
    assert np.all(got["cols"] > got["rows"])


@pytest.mark.parametrize("name", ["plate_24x18.mtx", "skewed_1500.mtx", "integer_30.mtx"])
def test_split_reassembles_to_scipy_matrix(name):
    sio = pytest.importorskip("scipy.io")
    sp = pytest.importorskip("scipy.sparse")
    path = os.path.join(MTX, name)
    rc, got = read_with_library(path)
    assert rc == 0
    n = got["n"]
    up = sp.coo_matrix((got["upper"], (got["rows"], got["cols"])), shape=(n, n))
    full = (sp.diags(got["diag"]) + up + up.T).tocsr()
    ref = sp.csr_matrix(sio.mmread(path))
    assert abs(full - ref).max() == 0.0


def test_general_drops_lower_and_last_diagonal_wins():
    rc, got = read_with_library(os.path.join(MTX, "general_60.mtx"))
    assert rc == 0 and not got["symmetric"]
    assert got["diag"][7] == 99.5 and got["diag"][8] == 58.0


def test_errors(tmp_path):
    def rc_of(text):
        p = tmp_path / "m.mtx"
        p.write_text(text)
        return read_with_library(str(p))[0]
    assert read_with_library(str(tmp_path / "absent.mtx"))[0] == -1
    assert rc_of("not a banner\n1 1 1\n1 1 1.0\n") == -2
    assert rc_of("%%MatrixMarket matrix array real general\n1 1\n1.0\n") == -2
    assert rc_of("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1.0 0.0\n") == -2
    assert rc_of("%%MatrixMarket matrix coordinate real symmetric\n% only comments\n") == -3
    assert rc_of("%%MatrixMarket matrix coordinate real general\n3 4 1\n1 1 1.0\n") == -3
    assert rc_of("%%MatrixMarket matrix coordinate real symmetric\n3 3 2\n1 1 1.0\n") == -4
    assert rc_of("%%MatrixMarket matrix coordinate real symmetric\n3 3 1\n4 1 1.0\n") == -4
    assert rc_of("%%MatrixMarket matrix coordinate real symmetric\n3 3 0\n") == 0
