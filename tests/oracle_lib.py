"""ctypes view of oracle/liboracle.so (the CPU restatement).  Checker only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
up = C.POINTER(C.c_uint64)

FMT_COO, FMT_CSR, FMT_ELL_UPRM, FMT_ELL_UPCM, FMT_ELL_FULL = range(5)


def D(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(dp)


def I(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(ip)


def load():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
    lib = C.CDLL(so)
    for f in ("orc_vec_dot", "orc_vec_l1norm", "orc_rmse", "orc_euclidean_distance", "orc_max_difference"):
        getattr(lib, f).restype = C.c_double
    lib.orc_vec_axpy.argtypes = [dp, dp, C.c_double, dp, C.c_int]
    lib.orc_cg.argtypes = [C.c_int, C.c_int, dp, C.c_int, dp, ip, ip, C.c_int, dp, dp, C.c_int, C.c_double, dp]
    lib.orc_cg.restype = C.c_int
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def max_nnz_full(n, rows, cols):
    return lib().orc_compute_max_nnz_row_full(n, rows.size, I(rows), I(cols))


def max_nnz_upper(n, rows, cols):
    return lib().orc_compute_max_nnz_row_upper(n, rows.size, I(rows), I(cols))


def coo_to_ell_full(n, upper, rows, cols, K):
    ev = np.empty(n * K)
    ec = np.empty(n * K, dtype=np.int32)
    st = lib().orc_coo_to_ell_symmetric_full_colmajor(n, rows.size, D(upper), I(rows), I(cols), D(ev), I(ec), K)
    return ev, ec, st


def coo_to_ell_upper_colmajor(n, upper, rows, cols, K):
    ev = np.empty(n * K)
    ec = np.empty(n * K, dtype=np.int32)
    st = lib().orc_coo_to_ell_symmetric_upper_colmajor(n, rows.size, D(upper), I(rows), I(cols), D(ev), I(ec), K)
    return ev, ec, st


def coo_to_ell_upper_rowmajor(n, upper, rows, cols, K):
    ev = np.empty(n * K)
    ec = np.empty(n * K, dtype=np.int32)
    st = lib().orc_coo_to_ell_symmetric_upper(n, rows.size, D(upper), I(rows), I(cols), D(ev), I(ec), K)
    return ev, ec, st


def coo_to_csr(n, rows):
    rp = np.empty(n + 1, dtype=np.int32)
    lib().orc_coo_to_csr(n, rows.size, I(rows), I(rp))
    return rp


def mv_ell_full(n, K, diag, ev, ec, x):
    y = np.empty(n)
    lib().orc_mv_ell_symmetric_full_colmajor(n, K, D(diag), D(ev), I(ec), D(x), D(y))
    return y


def mv_coo(n, diag, upper, rows, cols, x):
    y = np.empty(n)
    lib().orc_mv_coo_symmetric(n, rows.size, D(diag), D(upper), I(rows), I(cols), D(x), D(y))
    return y


def mv_csr(n, diag, upper, cols, row_ptr, x):
    y = np.empty(n)
    lib().orc_mv_csr_symmetric(n, D(diag), D(upper), I(cols), I(row_ptr), D(x), D(y))
    return y


def mv_ell_upper_colmajor(n, K, diag, ev, ec, x):
    y = np.empty(n)
    st = lib().orc_mv_ell_symmetric_upper_colmajor(n, K, D(diag), D(ev), I(ec), D(x), D(y))
    return y, st


def mv_ell_upper_rowmajor(n, K, diag, ev, ec, x):
    y = np.empty(n)
    st = lib().orc_mv_ell_symmetric_upper(n, K, D(diag), D(ev), I(ec), D(x), D(y))
    return y, st


def cg(fmt, n, diag, count, vals, ia, ib, K, b, max_iter, tol):
    """-> (ret, x, traj[iters+1, 4])"""
    x = np.full(n, 3.0)
    traj = np.full((max(max_iter, 1), 4), np.nan)
    null = C.cast(None, ip)
    ret = lib().orc_cg(fmt, n, D(diag), count, D(vals), I(ia), I(ib) if ib is not None else null,
                       K, D(b), D(x), max_iter, tol, traj.ctypes.data_as(dp))
    done = (ret - 1) if ret > 0 else (max_iter - 1)
    return ret, x, traj[:max(done, 0) + 1]


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype == np.float64 else a
