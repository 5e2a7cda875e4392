"""GPU parity of the reference-signature entry points (include/cg.h, csr.h, ell.h, vec.h, coo.h) exported by
libcg_hip.so: called exactly as the reference's own tests call them (host pointers), compared with the
reference goldens / the oracle.  Bit-exact for every index array, every SpMV and every strict-mode solve."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import foamio
import oracle_lib as O
from gpu_common import GOLDENS, bits, golden, inputs

pytestmark = pytest.mark.gpu

dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint64)
D, I = O.D, O.I


@pytest.fixture(scope="module")
def api():
    os.environ["CG_HIP_QUIET"] = "1"
    import conjugate_gradient_amd as cga
    L = cga.lib()
    for f in ("vec_dot", "vec_l1norm", "rmse", "euclidean_distance", "max_difference"):
        getattr(L, f).restype = C.c_double
    L.vec_axpy.argtypes = [dp, dp, C.c_double, dp, C.c_int]
    for f in ("conjugate_gradient_coo", "conjugate_gradient_csr"):
        getattr(L, f).argtypes = [C.c_int, dp, C.c_int, dp, ip, ip, dp, dp, C.c_int, C.c_double]
    for f in ("conjugate_gradient_ell", "conjugate_gradient_ell_colmajor", "conjugate_gradient_ell_full_colmajor"):
        getattr(L, f).argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
    return L


@pytest.mark.parametrize("tag", GOLDENS)
def test_conversions_bit_exact(api, tag):
    s, g = inputs(tag), golden(tag)
    n, U = s["n"], s["U"]
    Kf, Ku = int(g["K_full"]), int(g["K_upper"])
    assert api.compute_max_nnz_row_full(n, U, I(s["rows"]), I(s["cols"])) == Kf
    assert api.compute_max_nnz_row_upper(n, U, I(s["rows"]), I(s["cols"])) == Ku
    ev = np.full(n * Kf, 7.0); ec = np.full(n * Kf, 7, dtype=np.int32)
    api.coo_to_ell_symmetric_full_colmajor(n, U, D(s["upper"]), I(s["rows"]), I(s["cols"]), D(ev), I(ec), Kf)
    assert np.array_equal(ec, g["ell_full_cols"]) and np.array_equal(bits(ev), bits(g["ell_full_vals"]))
    ec64 = np.full(n * Kf, 7, dtype=np.uint64)
    api.coo_to_ell_symmetric_full_colmajor_sdtint(n, U, D(s["upper"]), I(s["rows"]), I(s["cols"]), D(ev),
                                                  ec64.ctypes.data_as(up), Kf)
    assert np.array_equal(ec64, g["ell_full_cols"].astype(np.int64).astype(np.uint64))
    uv = np.full(n * Ku, 7.0); uc = np.full(n * Ku, 7, dtype=np.int32)
    api.coo_to_ell_symmetric_upper_colmajor(n, U, D(s["upper"]), I(s["rows"]), I(s["cols"]), D(uv), I(uc), Ku)
    assert np.array_equal(uc, g["ell_upcm_cols"]) and np.array_equal(bits(uv), bits(g["ell_upcm_vals"]))
    api.coo_to_ell_symmetric_upper(n, U, D(s["upper"]), I(s["rows"]), I(s["cols"]), D(uv), I(uc), Ku)
    assert np.array_equal(uc, g["ell_uprm_cols"]) and np.array_equal(bits(uv), bits(g["ell_uprm_vals"]))
    rp = np.full(n + 1, 7, dtype=np.int32)
    assert api.coo_to_csr(n, U, D(s["upper"]), I(s["rows"]), I(s["cols"]), I(rp)) == 0
    assert np.array_equal(rp, g["row_ptr"])


def _random_upper(rng, n, U):
    r = rng.integers(0, n - 1, U); c = rng.integers(0, n, U)
    keep = c > r
    pairs = np.unique(np.stack([r[keep], c[keep]], 1), axis=0)
    rng.shuffle(pairs)                       # NOT row-sorted: slot order must follow entry order
    return pairs[:, 0].astype(np.int32).copy(), pairs[:, 1].astype(np.int32).copy()


def test_conversion_unsorted_empty_and_error_paths(api):
    rng = np.random.default_rng(17)
    n = 700
    rows, cols = _random_upper(rng, n, 4000)
    U = rows.size
    vals = rng.standard_normal(U)
    K = O.max_nnz_full(n, rows, cols)
    assert api.compute_max_nnz_row_full(n, U, I(rows), I(cols)) == K
    want_v, want_c, st = O.coo_to_ell_full(n, vals, rows, cols, K)
    ev = np.empty(n * K); ec = np.empty(n * K, dtype=np.int32)
    api.coo_to_ell_symmetric_full_colmajor(n, U, D(vals), I(rows), I(cols), D(ev), I(ec), K)
    assert st == 0 and np.array_equal(ec, want_c) and np.array_equal(bits(ev), bits(want_v))
    # K too small: the reference stops at the first overflowing placement and keeps what it wrote (ell.c:834-852)
    K2 = max(K - 2, 1)
    want_v, want_c, st = O.coo_to_ell_full(n, vals, rows, cols, K2)
    ev = np.empty(n * K2); ec = np.empty(n * K2, dtype=np.int32)
    api.coo_to_ell_symmetric_full_colmajor(n, U, D(vals), I(rows), I(cols), D(ev), I(ec), K2)
    assert st == 1 and np.array_equal(ec, want_c) and np.array_equal(bits(ev), bits(want_v))
    # a non-upper entry in the middle: -1 from the counters, partial output from the converter (ell.c:823-828)
    bad_r, bad_c = rows.copy(), cols.copy()
    bad_r[U // 2], bad_c[U // 2] = bad_c[U // 2], bad_r[U // 2]
    assert api.compute_max_nnz_row_full(n, U, I(bad_r), I(bad_c)) == -1
    assert api.compute_max_nnz_row_upper(n, U, I(bad_r), I(bad_c)) == -1
    want_v, want_c, st = O.coo_to_ell_full(n, vals, bad_r, bad_c, K)
    ev = np.empty(n * K); ec = np.empty(n * K, dtype=np.int32)
    api.coo_to_ell_symmetric_full_colmajor(n, U, D(vals), I(bad_r), I(bad_c), D(ev), I(ec), K)
    assert st == 1 and np.array_equal(ec, want_c) and np.array_equal(bits(ev), bits(want_v))
    # empty matrix (diag only)
    e = np.zeros(0, dtype=np.int32)
    assert api.compute_max_nnz_row_full(5, 0, I(e), I(e)) == 0
    rp = np.full(6, 9, dtype=np.int32)
    api.coo_to_csr(5, 0, D(np.zeros(0)), I(e), I(e), I(rp))
    assert np.array_equal(rp, O.coo_to_csr(5, e))
    # coo_to_csr with gaps and with unsorted rows (running-maximum semantics of the reference's while loop)
    for rr in (np.array([0, 0, 2, 2, 5, 9], dtype=np.int32), np.array([3, 1, 1, 6, 2, 8, 8], dtype=np.int32)):
        rp = np.full(12, 9, dtype=np.int32)
        api.coo_to_csr(11, rr.size, D(np.zeros(rr.size)), I(rr), I(rr), I(rp))
        assert np.array_equal(rp, O.coo_to_csr(11, rr))


def test_sorting_coo_to_csr(api):
    rng = np.random.default_rng(5)
    n, nnz = 300, 5000
    rows = rng.integers(0, n, nnz).astype(np.int32); cols = rng.integers(0, n, nnz).astype(np.int32)
    vals = rng.standard_normal(nnz)
    rp = np.empty(n + 1, dtype=np.int32); cc = np.empty(nnz, dtype=np.int32); cv = np.empty(nnz)
    api.sorting_coo_to_csr(n, nnz, D(vals), I(rows), I(cols), I(rp), I(cc), D(cv))
    wrp = np.empty(n + 1, dtype=np.int32); wcc = np.empty(nnz, dtype=np.int32); wcv = np.empty(nnz)
    O.lib().orc_sorting_coo_to_csr(n, nnz, D(vals), I(rows), I(cols), I(wrp), I(wcc), D(wcv))
    assert np.array_equal(rp, wrp) and np.array_equal(cc, wcc) and np.array_equal(bits(cv), bits(wcv))


@pytest.mark.parametrize("tag", GOLDENS)
def test_spmv_all_formats_bit_exact(api, tag):
    s, g = inputs(tag), golden(tag)
    n, U, Kf, Ku = s["n"], s["U"], int(g["K_full"]), int(g["K_upper"])
    x = np.ascontiguousarray(g["x_probe"])
    evf, ecf = np.ascontiguousarray(g["ell_full_vals"]), np.ascontiguousarray(g["ell_full_cols"])
    y = np.full(n, 7.0)
    api.mv_ell_symmetric_full_colmajor(n, Kf, D(s["diag"]), D(evf), I(ecf), D(x), D(y))
    assert np.array_equal(bits(y), bits(g["y_probe"]))
    ec64 = ecf.astype(np.int64).astype(np.uint64)
    y[:] = 7
    api.mv_ell_symmetric_full_colmajor_sdtint(n, Kf, D(s["diag"]), D(evf), ec64.ctypes.data_as(up), D(x), D(y))
    assert np.array_equal(bits(y), bits(g["y_probe"]))
    y[:] = 7
    api.mv_coo_symmetric(n, U, D(s["diag"]), D(s["upper"]), I(s["rows"]), I(s["cols"]), D(x), D(y))
    assert np.array_equal(bits(y), bits(g["y_probe_coo"]))
    y[:] = 7
    rp = np.ascontiguousarray(g["row_ptr"])
    api.mv_csr_symmetric(n, D(s["diag"]), D(s["upper"]), I(s["cols"]), I(rp), D(x), D(y))
    assert np.array_equal(bits(y), bits(g["y_probe"]))
    y[:] = 7
    uv, uc = np.ascontiguousarray(g["ell_upcm_vals"]), np.ascontiguousarray(g["ell_upcm_cols"])
    api.mv_ell_symmetric_upper_colmajor(n, Ku, D(s["diag"]), D(uv), I(uc), D(x), D(y))
    assert np.array_equal(bits(y), bits(g["y_probe_upcm"]))
    rv, rc = np.ascontiguousarray(g["ell_uprm_vals"]), np.ascontiguousarray(g["ell_uprm_cols"])
    want, st = O.mv_ell_upper_rowmajor(n, Ku, s["diag"], rv, rc, x)
    for f in (api.mv_ell_symmetric_upper, api.mv_ell_symmetric_upper_opt):
        y[:] = 7
        f(n, Ku, D(s["diag"]), D(rv), I(rc), D(x), D(y))
        assert st == 0 and np.array_equal(bits(y), bits(want))
    # generic row-major ELL (no diagonal): feed it the full rows transposed
    rm_v = np.ascontiguousarray(evf.reshape(Kf, n).T); rm_c = np.ascontiguousarray(ecf.reshape(Kf, n).T)
    want = np.empty(n)
    O.lib().orc_mv_ell(n, Kf, D(rm_v), I(rm_c), D(x), D(want))
    y[:] = 7
    api.mv_ell(n, Kf, D(rm_v), I(rm_c), D(x), D(y))
    assert np.array_equal(bits(y), bits(want))


@pytest.mark.parametrize("tag", ["cylinder_2000", "poisson27_12x10x8"])
def test_upper_ell_spmv_partial_output_on_malformed_entry(api, tag):
    """a stored column <= its row: the reference prints an error and returns with y as accumulated so far
    (ell.c:182-210 row-major, :971-998 slot-major); the device path reproduces that partial y bit for bit"""
    s, g = inputs(tag), golden(tag)
    n, Ku = s["n"], int(g["K_upper"])
    x = np.ascontiguousarray(g["x_probe"])
    for vals_key, cols_key, oracle_fn, fn, row_major in (
            ("ell_upcm_vals", "ell_upcm_cols", O.mv_ell_upper_colmajor, api.mv_ell_symmetric_upper_colmajor, False),
            ("ell_uprm_vals", "ell_uprm_cols", O.mv_ell_upper_rowmajor, api.mv_ell_symmetric_upper, True)):
        v = np.ascontiguousarray(g[vals_key]); c = np.ascontiguousarray(g[cols_key]).copy()
        used = np.flatnonzero(c != -1)
        for where in (used[len(used) // 2], used[0], used[-1]):
            cc = c.copy()
            row = where // Ku if row_major else where % n
            cc[where] = row if where != used[0] else max(row - 1, 0)      # col == row, or col < row
            want, st = oracle_fn(n, Ku, s["diag"], v, cc, x)
            assert st != 0
            y = np.full(n, 7.0)
            fn(n, Ku, D(s["diag"]), D(v), I(cc), D(x), D(y))
            assert np.array_equal(bits(y), bits(want)), (vals_key, int(where))


@pytest.mark.parametrize("tag", ["cylinder_2000", "cylinder_8000", "poisson7_16", "poisson27_12x10x8"])
def test_cg_solvers_bit_exact(api, tag):
    s, g = inputs(tag), golden(tag)
    n, U, Kf, Ku = s["n"], s["U"], int(g["K_full"]), int(g["K_upper"])
    mi, tol = int(g["max_iter"]), float(g["tol"])
    b = s["source"]
    rp = np.ascontiguousarray(g["row_ptr"])
    calls = [
        (O.FMT_COO, lambda x: api.conjugate_gradient_coo(n, D(s["diag"]), U, D(s["upper"]), I(s["rows"]), I(s["cols"]), D(b), D(x), mi, tol),
         (s["upper"], s["rows"], s["cols"], 0)),
        (O.FMT_CSR, lambda x: api.conjugate_gradient_csr(n, D(s["diag"]), U, D(s["upper"]), I(rp), I(s["cols"]), D(b), D(x), mi, tol),
         (s["upper"], rp, s["cols"], 0)),
        (O.FMT_ELL_UPRM, lambda x: api.conjugate_gradient_ell(n, D(s["diag"]), U, D(np.ascontiguousarray(g["ell_uprm_vals"])), I(np.ascontiguousarray(g["ell_uprm_cols"])), Ku, D(b), D(x), mi, tol),
         (np.ascontiguousarray(g["ell_uprm_vals"]), np.ascontiguousarray(g["ell_uprm_cols"]), None, Ku)),
        (O.FMT_ELL_UPCM, lambda x: api.conjugate_gradient_ell_colmajor(n, D(s["diag"]), U, D(np.ascontiguousarray(g["ell_upcm_vals"])), I(np.ascontiguousarray(g["ell_upcm_cols"])), Ku, D(b), D(x), mi, tol),
         (np.ascontiguousarray(g["ell_upcm_vals"]), np.ascontiguousarray(g["ell_upcm_cols"]), None, Ku)),
        (O.FMT_ELL_FULL, lambda x: api.conjugate_gradient_ell_full_colmajor(n, D(s["diag"]), U, D(np.ascontiguousarray(g["ell_full_vals"])), I(np.ascontiguousarray(g["ell_full_cols"])), Kf, D(b), D(x), mi, tol),
         (np.ascontiguousarray(g["ell_full_vals"]), np.ascontiguousarray(g["ell_full_cols"]), None, Kf)),
    ]
    for fmt, call, (vals, ia, ib, K) in calls:
        x = np.full(n, 5.0)
        ret = call(x)
        assert ret == int(g["cg_ret"][fmt]), (tag, fmt, ret)
        oret, ox, _ = O.cg(fmt, n, s["diag"], U, vals, ia, ib, K, b, mi, tol)
        assert oret == ret and np.array_equal(bits(x), bits(ox)), (tag, fmt)


def test_vec_functions(api):
    rng = np.random.default_rng(2)
    for n in (1, 7, 2048, 2049, 50001):
        a = rng.standard_normal(n); b = rng.standard_normal(n); out = np.empty(n); want = np.empty(n)
        assert api.vec_dot(D(a), D(b), n) == O.lib().orc_vec_dot(D(a), D(b), n)
        assert api.vec_l1norm(D(a), n) == O.lib().orc_vec_l1norm(D(a), n)
        for f, g_ in ((api.vec_sub, O.lib().orc_vec_sub), (api.vec_add, O.lib().orc_vec_add)):
            f(D(a), D(b), D(out), n); g_(D(a), D(b), D(want), n)
            assert np.array_equal(bits(out), bits(want))
        api.vec_axpy(D(a), D(b), 0.3, D(out), n); O.lib().orc_vec_axpy(D(a), D(b), 0.3, D(want), n)
        assert np.array_equal(bits(out), bits(want))
        a2 = a.copy(); api.vec_axpy(D(a2), D(b), -1.7, D(a2), n)       # out aliases a
        O.lib().orc_vec_axpy(D(a), D(b), -1.7, D(want), n)
        assert np.array_equal(bits(a2), bits(want))
        api.vec_assign(D(out), D(b), n)
        assert np.array_equal(bits(out), bits(b))
        assert api.rmse(D(a), D(b), n) == O.lib().orc_rmse(D(a), D(b), n)
        assert api.euclidean_distance(D(a), D(b), n) == O.lib().orc_euclidean_distance(D(a), D(b), n)
        assert api.max_difference(D(a), D(b), n) == O.lib().orc_max_difference(D(a), D(b), n)


_STDOUT_SNIPPET = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import foamio, conjugate_gradient_amd as cga
from gpu_common import inputs, golden
tag, which = %(tag)r, %(which)r
s, g = inputs(tag), golden(tag)
L = cga.lib()
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
D = lambda a: np.ascontiguousarray(a).ctypes.data_as(dp)
I = lambda a: np.ascontiguousarray(a).ctypes.data_as(ip)
n, U = s["n"], s["U"]; x = np.zeros(n)
keep = [np.ascontiguousarray(g[k]) for k in ("ell_full_vals", "ell_full_cols", "row_ptr")]
if which == "ell_full":
    L.conjugate_gradient_ell_full_colmajor.argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
    L.conjugate_gradient_ell_full_colmajor(n, D(s["diag"]), U, D(keep[0]), I(keep[1]), int(g["K_full"]), D(s["source"]), D(x), %(mi)d, 1e-6)
elif which == "coo":
    L.conjugate_gradient_coo.argtypes = [C.c_int, dp, C.c_int, dp, ip, ip, dp, dp, C.c_int, C.c_double]
    L.conjugate_gradient_coo(n, D(s["diag"]), U, D(s["upper"]), I(s["rows"]), I(s["cols"]), D(s["source"]), D(x), %(mi)d, 1e-6)
elif which == "csr":
    L.conjugate_gradient_csr.argtypes = [C.c_int, dp, C.c_int, dp, ip, ip, dp, dp, C.c_int, C.c_double]
    L.conjugate_gradient_csr(n, D(s["diag"]), U, D(s["upper"]), I(keep[2]), I(s["cols"]), D(s["source"]), D(x), %(mi)d, 1e-6)
else:
    rp = np.zeros(n + 1, dtype=np.int32)
    L.coo_to_csr(n, U, D(s["upper"]), I(s["rows"]), I(s["cols"]), rp.ctypes.data_as(ip))
"""


@pytest.mark.parametrize("tag,mi", [("cylinder_2000", 1000), ("poisson27_12x10x8", 400)])
@pytest.mark.parametrize("which", ["ell_full", "coo", "csr", "coo_to_csr"])
def test_stdout_text_matches_reference(tag, mi, which):
    """The solvers' stdout (banner, r[0..4], one residual line per iteration, final line) is part of the
    reference's behaviour (cg.c:426-501); compare with the text the reference printed (tests/golden/stdout_*)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _STDOUT_SNIPPET % dict(root=root, tests=os.path.join(root, "tests"), tag=tag, which=which, mi=mi)
    env = dict(os.environ)
    env.pop("CG_HIP_QUIET", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    with open(os.path.join(foamio.GOLDEN, "stdout_%s_%s.txt" % (tag, which))) as f:
        want = f.read()
    assert out.stdout == want


@pytest.mark.parametrize("fmt", ["ell", "coo", "csr"])
def test_c_host_driver_end_to_end(fmt):
    """conjugate_gradient_amd/host/test_cg_hip: the reference's `make test_cg` flow (tests/tests_cg.c:1061-1299) as a C
    program over libcg_hip: file -> convert -> CG on the GPU -> compare with OpenFOAM's Phi (reference data/actual.txt:257-274)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "conjugate_gradient_amd", "host", "test_cg_hip")
    assert os.path.exists(exe), "build it with `make -C conjugate_gradient_amd/host` (done by __graft_entry__.build())"
    env = dict(os.environ)
    env.pop("CG_HIP_QUIET", None)
    out = subprocess.run([exe, os.path.join(foamio.GOLDEN, "cylinder_2000.system"), os.path.join(foamio.GOLDEN, "Phi_2000"),
                          "--format", fmt], capture_output=True, text=True, env=env, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stdout
    assert "Converged in 254 iterations." in text
    assert "Iteration 252: Residual norm = 1.08951e-06" in text
    assert "RMSE: 0.003061" in text and "Max component difference: 0.004555" in text
    if fmt == "ell":
        assert "nnz_max = 4" in text and "padding = 220 " in text          # 8000 slots, 220 pads


def _coo_solve(api, n, diag, upper, rows, cols, b, max_iter=1000, tol=1e-6):
    x = np.full(n, 5.0)
    e = np.zeros(0, dtype=np.int32)
    ret = api.conjugate_gradient_coo(n, D(diag), rows.size, D(upper) if upper.size else None,
                                     I(rows) if rows.size else None, I(cols) if cols.size else None, D(b), D(x), max_iter, tol)
    oret, ox, traj = O.cg(O.FMT_COO, n, diag, rows.size, upper if upper.size else np.zeros(1),
                          rows if rows.size else np.zeros(1, dtype=np.int32), cols if cols.size else np.zeros(1, dtype=np.int32),
                          0, b, max_iter, tol)
    return ret, x, oret, ox


def test_reference_toy_cases(api):
    """the reference's toy drivers (tests/tests_cg.c:14-249): zero matrix, identity, dense uniform 1/n -- they only print
    there; here the oracle supplies the expected return value and x"""
    n = 2000
    e = np.zeros(0, dtype=np.int32); ev = np.zeros(0)
    b = np.arange(1, n + 1, dtype=np.float64)
    # zero matrix with NULL upper/rows/cols (tests_cg.c:20-46): p.Ap = 0 -> alpha = inf, NaNs, never converges
    ret, x, oret, ox = _coo_solve(api, n, np.zeros(n), ev, e, e, b)
    assert ret == oret == -1 and np.array_equal(np.isnan(x), np.isnan(ox)) and np.isnan(x).all()
    # identity (tests_cg.c:89-121): one step
    ret, x, oret, ox = _coo_solve(api, n, np.ones(n), ev, e, e, b)
    assert ret == oret == 2 and np.array_equal(bits(x), bits(ox)) and np.array_equal(x, b)
    # dense uniform 1/n (tests_cg.c:163-212), smaller n to keep the test quick: rank-1 matrix, row width n-1
    m = 300
    iu = np.triu_indices(m, 1)
    rows = iu[0].astype(np.int32); cols = iu[1].astype(np.int32)
    ret, x, oret, ox = _coo_solve(api, m, np.full(m, 1.0 / m), np.full(rows.size, 1.0 / m), rows, cols, np.ones(m))
    assert ret == oret and np.array_equal(bits(x), bits(ox))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 64, 65, 511])
def test_tiny_and_odd_sizes(api, n):
    rng = np.random.default_rng(n)
    rows, cols = (np.zeros(0, dtype=np.int32),) * 2
    if n > 1:
        iu = np.triu_indices(n, 1)
        keep = rng.random(iu[0].size) < min(1.0, 6.0 / n)
        rows = iu[0][keep].astype(np.int32); cols = iu[1][keep].astype(np.int32)
    upper = -rng.random(rows.size)
    diag = np.full(n, 1.0)
    np.add.at(diag, rows, -upper); np.add.at(diag, cols, -upper)
    b = rng.standard_normal(n)
    for mi, tol in ((1000, 1e-6), (1, 1e-6), (2, 1e-6), (3, 0.0), (50, 1e300)):
        ret, x, oret, ox = _coo_solve(api, n, diag, upper, rows, cols, b, mi, tol)
        assert ret == oret, (n, mi, tol, ret, oret)
        assert np.array_equal(bits(x), bits(ox)), (n, mi, tol)


def test_coo_and_csr_take_any_entries_like_the_reference(api):
    """The reference's COO and CSR SpMVs and solvers scatter whatever (i, j) they are given (coo.c:202-233, csr.c:190-229,
    cg.c:21-222): lower-triangle entries, unsorted lists, duplicates, and (i, i) entries, which then count twice.  Same
    bits here (the upper-triangle requirement belongs to the ELL converters only)."""
    rng = np.random.default_rng(31)
    n, cnt = 777, 5000
    rows = rng.integers(0, n, cnt).astype(np.int32)
    cols = rng.integers(0, n, cnt).astype(np.int32)
    rows[:40] = cols[:40]                                   # entries on the diagonal
    rows[40:60], cols[40:60] = rows[60:80], cols[60:80]     # duplicates
    assert (rows > cols).any() and (rows < cols).any()
    vals = -rng.random(cnt)
    diag = np.full(n, 1.0)
    np.add.at(diag, rows, -vals); np.add.at(diag, cols, -vals)   # dominant whatever the list holds
    x = rng.standard_normal(n)
    y = np.full(n, 7.0)
    api.mv_coo_symmetric(n, cnt, D(diag), D(vals), I(rows), I(cols), D(x), D(y))
    assert np.array_equal(bits(y), bits(O.mv_coo(n, diag, vals, rows, cols, x)))
    # the same list as CSR rows (any column in any row)
    order = np.argsort(rows, kind="stable")
    r2, c2, v2 = rows[order], np.ascontiguousarray(cols[order]), np.ascontiguousarray(vals[order])
    rp = np.zeros(n + 1, dtype=np.int32); np.add.at(rp, r2 + 1, 1); rp = np.cumsum(rp).astype(np.int32)
    y[:] = 7.0
    api.mv_csr_symmetric(n, D(diag), D(v2), I(c2), I(rp), D(x), D(y))
    assert np.array_equal(bits(y), bits(O.mv_csr(n, diag, v2, c2, rp, x)))
    b = rng.standard_normal(n)
    ret, xs, oret, ox = _coo_solve(api, n, diag, vals, rows, cols, b)
    assert ret == oret and ret > 0 and np.array_equal(bits(xs), bits(ox))
    xs = np.full(n, 5.0)
    ret = api.conjugate_gradient_csr(n, D(diag), cnt, D(v2), I(rp), I(c2), D(b), D(xs), 1000, 1e-6)
    oret, ox, _ = O.cg(O.FMT_CSR, n, diag, cnt, v2, rp, c2, 0, b, 1000, 1e-6)
    assert ret == oret and np.array_equal(bits(xs), bits(ox))
    # an index outside the matrix stays an error (the reference would write out of bounds): -1, nothing solved
    bad = cols.copy(); bad[7] = n
    xs = np.full(n, 5.0)
    assert api.conjugate_gradient_coo(n, D(diag), cnt, D(vals), I(rows), I(bad), D(b), D(xs), 1000, 1e-6) == -1


def test_huge_max_iter_is_served_in_bounded_memory():
    """max_iter = INT_MAX ("until converged"): the reference's loop is O(1) in max_iter (src/cg.c:414-508) and so is the
    drop-in -- the history lives in a bounded window -- so the call returns the reference's 254 on foam-2000."""
    import ctypes as C
    import conjugate_gradient_amd as cga
    L = cga.lib()
    s, g = inputs("cylinder_2000"), golden("cylinder_2000")
    n, K = s["n"], int(g["K_full"])
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.conjugate_gradient_ell_full_colmajor.argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
    ev = np.ascontiguousarray(g["ell_full_vals"]); ec = np.ascontiguousarray(g["ell_full_cols"])
    diag = np.ascontiguousarray(s["diag"]); b = np.ascontiguousarray(s["source"]); x = np.full(n, 5.0)
    ret = L.conjugate_gradient_ell_full_colmajor(n, diag.ctypes.data_as(dp), s["U"], ev.ctypes.data_as(dp), ec.ctypes.data_as(ip), K,
                                                 b.ctypes.data_as(dp), x.ctypes.data_as(dp), 2 ** 31 - 1, float(g["tol"]))
    assert ret == int(g["cg_ret"][4]) == 254
    assert np.array_equal(x.view(np.uint64), np.ascontiguousarray(g["cg_x"]).view(np.uint64))
