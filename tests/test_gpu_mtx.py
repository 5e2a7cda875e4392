"""GPU parity on MatrixMarket inputs: file -> mm_read_symmetric_split (include/mm_reader.h) -> the
reference-signature conversions / SpMVs / solvers of libcg_hip, as reference tests/test_vec.c:443-633 strings them
together, against the oracle on the same arrays.  Bit-exact index arrays, SpMV and strict CG trajectory; the skewed
matrix also goes through the CSR stream/adaptive and sliced-ELL layouts of the handle API."""
import ctypes as C
import os

import numpy as np
import pytest

import foamio
import oracle_lib as O
from gpu_common import bits, same_bits_nan
from test_mm_reader import MTX, read_with_library

pytestmark = pytest.mark.gpu
dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint64)
D, I = O.D, O.I


@pytest.fixture(scope="module")
def cga():
    os.environ["CG_HIP_QUIET"] = "1"
    import conjugate_gradient_amd as m
    m.init(0)
    L = m.lib()
    L.conjugate_gradient_ell_full_colmajor.argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
    L.conjugate_gradient_coo.argtypes = [C.c_int, dp, C.c_int, dp, ip, ip, dp, dp, C.c_int, C.c_double]
    return m


@pytest.mark.parametrize("name", ["plate_24x18.mtx", "skewed_1500.mtx", "integer_30.mtx", "pattern_40.mtx"])
def test_mtx_pipeline_bit_exact(cga, name):
    L = cga.lib()
    rc, m = read_with_library(os.path.join(MTX, name))
    assert rc == 0
    n, U = m["n"], m["U"]
    rng = np.random.default_rng(1)
    xh = rng.random(n)
    # test_vec.c:572-581: K, full slot-major ELL
    K = L.compute_max_nnz_row_full(n, U, I(m["rows"]), I(m["cols"]))
    assert K == O.max_nnz_full(n, m["rows"], m["cols"])
    ev = np.empty(n * K); ec = np.empty(n * K, dtype=np.int32)
    L.coo_to_ell_symmetric_full_colmajor(n, U, D(m["upper"]), I(m["rows"]), I(m["cols"]), D(ev), I(ec), K)
    wv, wc, st = O.coo_to_ell_full(n, m["upper"], m["rows"], m["cols"], K)
    assert st == 0 and np.array_equal(ec, wc) and np.array_equal(bits(ev), bits(wv))
    # test_vec.c:594-600: 64-bit column twin
    y = np.zeros(n)
    ec64 = ec.astype(np.int64).astype(np.uint64)
    L.mv_ell_symmetric_full_colmajor_sdtint(n, K, D(m["diag"]), D(ev), ec64.ctypes.data_as(up), D(xh), D(y))
    want = O.mv_ell_full(n, K, m["diag"], ev, ec, xh)
    assert np.array_equal(bits(y), bits(want))
    y2 = np.zeros(n)
    L.mv_coo_symmetric(n, U, D(m["diag"]), D(m["upper"]), I(m["rows"]), I(m["cols"]), D(xh), D(y2))
    assert np.array_equal(bits(y2), bits(O.mv_coo(n, m["diag"], m["upper"], m["rows"], m["cols"], xh)))
    if name == "pattern_40.mtx":
        return   # not SPD (rows without a diagonal entry): conversions and SpMV only
    # CG, strict order, against the oracle's loop (these matrices are SPD by diagonal dominance)
    b = rng.random(n) - 0.5
    ret, xo, traj = O.cg(O.FMT_ELL_FULL, n, m["diag"], U, ev, ec, None, K, b, 400, 1e-8)
    x = np.full(n, 3.0)
    got = L.conjugate_gradient_ell_full_colmajor(n, D(m["diag"]), U, D(ev), I(ec), K, D(b), D(x), 400, 1e-8)
    assert got == ret and np.array_equal(bits(x), bits(xo))
    ret2, xo2, _ = O.cg(O.FMT_COO, n, m["diag"], U, m["upper"], m["rows"], m["cols"], 0, b, 400, 1e-8)
    x2 = np.full(n, 3.0)
    got2 = L.conjugate_gradient_coo(n, D(m["diag"]), U, D(m["upper"]), I(m["rows"]), I(m["cols"]), D(b), D(x2), 400, 1e-8)
    assert got2 == ret2 and np.array_equal(bits(x2), bits(xo2))


@pytest.mark.parametrize("layout", ["LAYOUT_CSR", "LAYOUT_SELL", "LAYOUT_ELL"])
def test_skewed_mtx_through_device_layouts(cga, layout):
    """rows of 1 .. several hundred entries through every device layout: SpMV bit-exact in the reference's per-row
    order (ELL, sliced ELL, CSR with exact rows); CSR's default uses a tree for rows > 64 entries: bit-exact on the
    short rows, 1e-13 relative on the long ones"""
    rc, m = read_with_library(os.path.join(MTX, "skewed_1500.mtx"))
    assert rc == 0
    n, U = m["n"], m["U"]
    A = cga.Matrix.from_coo(n, m["diag"], m["upper"], m["rows"], m["cols"], getattr(cga, layout))
    rng = np.random.default_rng(2)
    xh = rng.random(n)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    A.spmv(x, y)
    want = O.mv_coo(n, m["diag"], m["upper"], m["rows"], m["cols"], xh)
    got = y.to_host()
    counts = np.bincount(np.concatenate([m["rows"], m["cols"]]), minlength=n)
    short = counts <= 64
    assert counts.max() > 200 and short.sum() > 1000
    if layout == "LAYOUT_CSR":
        assert np.array_equal(bits(got[short]), bits(want[short]))
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
        A.set_exact_rows(True)
        A.spmv(x, y)
        assert np.array_equal(bits(y.to_host()), bits(want))
    else:
        assert np.array_equal(bits(got), bits(want))
