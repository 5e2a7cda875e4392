"""Pins the oracle (oracle/cg_oracle.c): (1) against every known-answer vector
the reference's own tests hold for this path, (2) against goldens produced by
the reference itself (oracle/gen_golden.py -> tests/golden/golden_*.npz).
CPU only."""
import os

import numpy as np
import pytest

import foamio
import oracle_lib as O
from oracle_lib import bits

GOLDENS = ["cylinder_2000", "cylinder_8000", "cylinder_32k",
           "poisson7_16", "poisson7_32x24x20", "poisson27_12x10x8"]


def _inputs(tag):
    if tag.startswith("cylinder_"):
        return foamio.foam(tag.split("_")[1])
    if tag == "poisson7_16":
        return foamio.poisson_coo(16, 16, 16, 0x5EED0001, 7)
    if tag == "poisson7_32x24x20":
        return foamio.poisson_coo(32, 24, 20, 0x5EED0001, 7)
    return foamio.poisson_coo(12, 10, 8, 0x5EED0001, 27)


def _golden(tag):
    return np.load(os.path.join(foamio.GOLDEN, "golden_%s.npz" % tag))


# ---- (1) the reference's own KATs ---------------------------------------

def test_kat_ell_5x5():
    # reference tests/test_ell.c:188-250: 5x5, y = {20,25,96,107,122} for all three ELL layouts
    diag = np.array([1., 4, 6, 9, 11])
    upper = np.array([2., 3, 5, 7, 8, 10])
    rows = np.array([0, 0, 1, 2, 2, 3], dtype=np.int32)
    cols = np.array([1, 4, 2, 3, 4, 4], dtype=np.int32)
    x = np.array([1., 2, 3, 4, 5])
    want = np.array([20., 25, 96, 107, 122])
    assert O.max_nnz_upper(5, rows, cols) == 2          # test_ell.c:30
    K = O.max_nnz_full(5, rows, cols)
    assert K == 3
    ev, ec, st = O.coo_to_ell_full(5, upper, rows, cols, K)
    assert st == 0
    assert np.array_equal(O.mv_ell_full(5, K, diag, ev, ec, x), want)
    uv, uc, st = O.coo_to_ell_upper_colmajor(5, upper, rows, cols, 2)
    y, st2 = O.mv_ell_upper_colmajor(5, 2, diag, uv, uc, x)
    assert st == 0 and st2 == 0 and np.array_equal(y, want)
    rv, rc, st = O.coo_to_ell_upper_rowmajor(5, upper, rows, cols, 2)
    y, st2 = O.mv_ell_upper_rowmajor(5, 2, diag, rv, rc, x)
    assert st == 0 and st2 == 0 and np.array_equal(y, want)
    # dense reconstruction (test_ell.c:197-202)
    A = np.diag(diag)
    for s in range(K):
        for i in range(5):
            if ec[s * 5 + i] >= 0:
                A[i, ec[s * 5 + i]] = ev[s * 5 + i]
    want_A = np.array([[1, 2, 0, 0, 3], [2, 4, 5, 0, 0], [0, 5, 6, 7, 8], [0, 0, 7, 9, 10], [3, 0, 8, 10, 11.]])
    assert np.array_equal(A, want_A)


def test_kat_csr_3x3():
    # reference tests/test_csr.c:77-99
    y = O.mv_csr(3, np.array([2., 3, 4]), np.array([1., 2, 3]), np.array([1, 2, 2], dtype=np.int32),
                 np.array([0, 2, 3, 3], dtype=np.int32), np.array([1., 2, 3]))
    assert np.array_equal(y, [10., 16, 20])


def test_kat_coo_csr_3x3():
    # reference tests/test_coo.c:143-176 (exact ==)
    diag = np.array([10., 20, 30]); up = np.array([1., 2, 3])
    rows = np.array([0, 0, 1], dtype=np.int32); cols = np.array([1, 2, 2], dtype=np.int32)
    v = np.array([1., 2, 3])
    assert np.array_equal(O.mv_coo(3, diag, up, rows, cols, v), [18., 50, 98])
    rp = O.coo_to_csr(3, rows)
    assert np.array_equal(O.mv_csr(3, diag, up, cols, rp, v), [18., 50, 98])


def test_kat_coo_to_csr_n4():
    # reference tests/test_csr.c:143-196: rows {0,0,2} -> row_ptr {0,2,2,3,3}
    rows = np.array([0, 0, 2], dtype=np.int32)
    assert np.array_equal(O.coo_to_csr(4, rows), [0, 2, 2, 3, 3])


def test_sorting_coo_to_csr_stable():
    rng = np.random.default_rng(5)
    n, nnz = 50, 400
    rows = rng.integers(0, n, nnz).astype(np.int32)
    cols = rng.integers(0, n, nnz).astype(np.int32)
    vals = rng.standard_normal(nnz)
    rp = np.empty(n + 1, dtype=np.int32); cc = np.empty(nnz, dtype=np.int32); cv = np.empty(nnz)
    O.lib().orc_sorting_coo_to_csr(n, nnz, O.D(vals), O.I(rows), O.I(cols), O.I(rp), O.I(cc), O.D(cv))
    perm = np.argsort(rows, kind="stable")
    assert np.array_equal(cc, cols[perm]) and np.array_equal(cv, vals[perm])
    assert np.array_equal(rp, np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]))


def test_non_upper_entry_is_rejected():
    rows = np.array([0, 2], dtype=np.int32); cols = np.array([1, 1], dtype=np.int32)
    assert O.max_nnz_full(3, rows, cols) == -1       # ell.c:377-382
    assert O.max_nnz_upper(3, rows, cols) == -1      # ell.c:333-340
    ev, ec, st = O.coo_to_ell_full(3, np.array([1., 2.]), rows, cols, 2)
    assert st == 1                                    # partial output: first entry placed, then stop
    assert ec.tolist() == [1, 0, -1, -1, -1, -1]


# ---- (2) goldens produced by the reference -------------------------------

@pytest.mark.parametrize("tag", GOLDENS)
def test_formats_bit_exact(tag):
    s, g = _inputs(tag), _golden(tag)
    n = s["n"]
    assert n == int(g["n"]) and s["U"] == int(g["U"])
    Kf = O.max_nnz_full(n, s["rows"], s["cols"]); Ku = O.max_nnz_upper(n, s["rows"], s["cols"])
    assert (Kf, Ku) == (int(g["K_full"]), int(g["K_upper"]))
    ev, ec, st = O.coo_to_ell_full(n, s["upper"], s["rows"], s["cols"], Kf)
    assert st == 0 and np.array_equal(ec, g["ell_full_cols"]) and np.array_equal(bits(ev), bits(g["ell_full_vals"]))
    uv, uc, st = O.coo_to_ell_upper_colmajor(n, s["upper"], s["rows"], s["cols"], Ku)
    assert st == 0 and np.array_equal(uc, g["ell_upcm_cols"]) and np.array_equal(bits(uv), bits(g["ell_upcm_vals"]))
    rv, rc, st = O.coo_to_ell_upper_rowmajor(n, s["upper"], s["rows"], s["cols"], Ku)
    assert st == 0 and np.array_equal(rc, g["ell_uprm_cols"]) and np.array_equal(bits(rv), bits(g["ell_uprm_vals"]))
    assert np.array_equal(O.coo_to_csr(n, s["rows"]), g["row_ptr"])


@pytest.mark.parametrize("tag", GOLDENS)
def test_spmv_bit_exact(tag):
    s, g = _inputs(tag), _golden(tag)
    n, Kf, Ku = s["n"], int(g["K_full"]), int(g["K_upper"])
    x = g["x_probe"]
    want = bits(g["y_probe"])
    assert np.array_equal(bits(O.mv_ell_full(n, Kf, s["diag"], g["ell_full_vals"], g["ell_full_cols"], x)), want)
    assert np.array_equal(bits(O.mv_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], x)), bits(g["y_probe_coo"]))
    assert np.array_equal(bits(O.mv_csr(n, s["diag"], s["upper"], s["cols"], g["row_ptr"], x)), want)
    y, st = O.mv_ell_upper_colmajor(n, Ku, s["diag"], g["ell_upcm_vals"], g["ell_upcm_cols"], x)
    assert st == 0 and np.array_equal(bits(y), bits(g["y_probe_upcm"]))


@pytest.mark.parametrize("tag", GOLDENS)
def test_cg_trajectory_bit_exact(tag):
    s, g = _inputs(tag), _golden(tag)
    n, Kf = s["n"], int(g["K_full"])
    ret, x, traj = O.cg(O.FMT_ELL_FULL, n, s["diag"], s["U"], g["ell_full_vals"], g["ell_full_cols"], None, Kf,
                        s["source"], int(g["max_iter"]), float(g["tol"]))
    assert ret == int(g["cg_ret"][4])
    assert np.array_equal(bits(x), bits(g["cg_x"]))
    gt = g["traj"]
    assert traj.shape == gt.shape
    assert np.array_equal(np.isnan(traj), np.isnan(gt))
    assert np.array_equal(bits(np.nan_to_num(traj)), bits(np.nan_to_num(gt)))


@pytest.mark.parametrize("tag", ["cylinder_2000", "cylinder_8000", "poisson27_12x10x8"])
def test_cg_all_formats_return_values(tag):
    s, g = _inputs(tag), _golden(tag)
    n, Ku = s["n"], int(g["K_upper"])
    mi, tol = int(g["max_iter"]), float(g["tol"])
    r0, x0, _ = O.cg(O.FMT_COO, n, s["diag"], s["U"], s["upper"], s["rows"], s["cols"], 0, s["source"], mi, tol)
    r1, x1, _ = O.cg(O.FMT_CSR, n, s["diag"], s["U"], s["upper"], g["row_ptr"], s["cols"], 0, s["source"], mi, tol)
    r2, _, _ = O.cg(O.FMT_ELL_UPRM, n, s["diag"], s["U"], g["ell_uprm_vals"], g["ell_uprm_cols"], None, Ku, s["source"], mi, tol)
    r3, _, _ = O.cg(O.FMT_ELL_UPCM, n, s["diag"], s["U"], g["ell_upcm_vals"], g["ell_upcm_cols"], None, Ku, s["source"], mi, tol)
    assert [r0, r1, r2, r3] == g["cg_ret"][:4].tolist()
    assert np.array_equal(bits(x0), bits(g["cg_x"])) and np.array_equal(bits(x1), bits(g["cg_x"]))


def test_foam2000_matches_reference_log():
    # reference data/actual.txt:257-274 and SURVEY §6: 254 returned, RMSE 0.003061, max diff 0.004555 vs data/Phi
    g = _golden("cylinder_2000")
    phi = foamio.read_foam_field(os.path.join(foamio.GOLDEN, "Phi_2000"))
    x = np.ascontiguousarray(g["cg_x"])
    assert int(g["cg_ret"][4]) == 254
    assert abs(O.lib().orc_rmse(O.D(x), O.D(phi), 2000) - 0.003061) < 5e-7
    assert abs(O.lib().orc_max_difference(O.D(x), O.D(phi), 2000) - 0.004555) < 5e-7


def test_openmp_variant_agrees_with_the_oracle_to_rounding():
    """oracle/cg_omp.c (bench.py's all-core CPU figure) runs the same loop with OpenMP reductions: same iteration count
    and solution as the order-exact oracle within rounding on a golden system"""
    import ctypes as C
    import os
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "libcg_omp.so")
    if not os.path.exists(so):
        pytest.skip("oracle/libcg_omp.so not built (run `make -C oracle`)")
    L = C.CDLL(so)
    s = foamio.poisson_coo(16, 16, 16, 0x5EED0001, 7)
    n = s["n"]
    K = O.max_nnz_full(n, s["rows"], s["cols"])
    ev, ec, st = O.coo_to_ell_full(n, s["upper"], s["rows"], s["cols"], K)
    ret, xo, traj = O.cg(O.FMT_ELL_FULL, n, s["diag"], 0, ev, ec, None, K, s["source"], 500, 1e-8)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.orc_cg_ell_full_omp.argtypes = [C.c_int, dp, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
    x = np.zeros(n)
    diag = np.ascontiguousarray(s["diag"]); b = np.ascontiguousarray(s["source"])
    got = L.orc_cg_ell_full_omp(n, diag.ctypes.data_as(dp), ev.ctypes.data_as(dp), ec.ctypes.data_as(ip), K, b.ctypes.data_as(dp),
                                x.ctypes.data_as(dp), 500, 1e-8)
    assert ret > 0 and abs(got - ret) <= 1
    assert np.max(np.abs(x - xo)) <= 1e-9 * max(1.0, np.max(np.abs(xo)))
    assert L.orc_omp_threads() >= 1
