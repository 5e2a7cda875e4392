"""CPU-only, build container only: the MatrixMarket inputs the REFERENCE holds (data/mkt_real_symmetric/bcsstk01, 04,
13 -- read by its tests/test_vec.c:443-600) through this repo's reader and through the path the reference feeds them to.
They are read where they lie under /root/reference (copying them into the repository was refused by the environment in
round 1), so nothing here can run on the GPU box: the device path sees MatrixMarket input only
through the synthetic fixtures of tests/golden/mtx (tests/test_gpu_mtx.py).  What this pins:
  * mm_read_symmetric_split == the restatement of the reference's inline reader == scipy's matrix, on real files;
  * the oracle (the checker of every GPU parity test) == the reference compiled from its sources, bit for bit, on
    real-world SPD matrices: compute_max_nnz_row_full, coo_to_ell_symmetric_full_colmajor, mv_coo_symmetric,
    mv_ell_symmetric_full_colmajor and the full-ELL CG solver (return value and x)."""
import ctypes as C
import os

import numpy as np
import pytest

import foamio
import oracle_lib as O
from test_mm_reader import read_with_library

REF_DATA = "/root/reference/data/mkt_real_symmetric"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
D = lambda a: a.ctypes.data_as(dp)  # noqa: E731
I = lambda a: a.ctypes.data_as(ip)  # noqa: E731
bits = lambda a: np.ascontiguousarray(a).view(np.uint64)  # noqa: E731


@pytest.mark.parametrize("name,n,U", [("bcsstk01.mtx", 48, 176), ("bcsstk04.mtx", 132, 1758), ("bcsstk13.mtx", 2003, 40940)])
def test_reference_held_mtx_files(name, n, U):
    path = os.path.join(REF_DATA, name)
    if not os.path.exists(path) or not os.path.exists(REF_SO):
        pytest.skip("reference data / libref.so not present (GPU box)")
    rc, got = read_with_library(path)
    assert rc == 0
    want = foamio.mtx_split(path)
    assert (got["n"], got["U"], got["symmetric"]) == (want["n"], want["U"], True) == (n, U, True)
    for k in ("diag", "upper", "rows", "cols"):
        assert np.array_equal(got[k], want[k]), k
    sio = pytest.importorskip("scipy.io")
    sp = pytest.importorskip("scipy.sparse")
    up = sp.coo_matrix((got["upper"], (got["rows"], got["cols"])), shape=(n, n))
    full = (sp.diags(got["diag"]) + up + up.T).tocsr()
    assert abs(full - sp.csr_matrix(sio.mmread(path))).max() == 0.0

    R = C.CDLL(REF_SO)
    diag, upper, rows, cols = got["diag"], got["upper"], got["rows"], got["cols"]
    K = R.compute_max_nnz_row_full(n, U, I(rows), I(cols))
    assert K == O.max_nnz_full(n, rows, cols) and K > 0
    ev = np.empty(n * K); ec = np.empty(n * K, dtype=np.int32)
    R.coo_to_ell_symmetric_full_colmajor(n, U, D(upper), I(rows), I(cols), D(ev), I(ec), K)
    oev, oec, st = O.coo_to_ell_full(n, upper, rows, cols, K)
    assert st == 0 and np.array_equal(ec, oec) and np.array_equal(bits(ev), bits(oev))
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    y_coo = np.empty(n); y_ell = np.empty(n)
    R.mv_coo_symmetric(n, U, D(diag), D(upper), I(rows), I(cols), D(x), D(y_coo))
    R.mv_ell_symmetric_full_colmajor(n, K, D(diag), D(ev), I(ec), D(x), D(y_ell))
    assert np.array_equal(bits(y_coo), bits(O.mv_coo(n, diag, upper, rows, cols, x)))
    assert np.array_equal(bits(y_ell), bits(O.mv_ell_full(n, K, diag, ev, ec, x)))
    # the solver: structural stiffness matrices, badly conditioned -- 300 iterations, whatever they reach
    b = full @ np.ones(n)
    xr = np.zeros(n)
    R.conjugate_gradient_ell_full_colmajor.argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
    libc = C.CDLL(None)
    libc.fflush(None)
    saved = os.dup(1); null = os.open(os.devnull, os.O_WRONLY); os.dup2(null, 1); os.close(null)
    try:
        ret = R.conjugate_gradient_ell_full_colmajor(n, D(diag), U, D(ev), I(ec), K, D(b), D(xr), 300, 1e-6)
        libc.fflush(None)
    finally:
        os.dup2(saved, 1); os.close(saved)
    oret, ox, traj = O.cg(O.FMT_ELL_FULL, n, diag, U, ev, ec, None, K, b, 300, 1e-6)
    assert ret == oret and np.array_equal(bits(xr), bits(ox))
