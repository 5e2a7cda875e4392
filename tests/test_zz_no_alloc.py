"""GPU: cgh_cg_solve_noalloc -- the one entry point that allocates nothing (workspace in a static __device__ array) --
against cgh_cg_solve, the entry point it is a variant of: same kernels, same launch shapes, so the residual history, the
solution and the converging iteration are bit-identical.  Two calls in a row, the second on a smaller system (the array
keeps the first call's contents); then the sizes it must refuse.  No reference counterpart."""
import numpy as np
import pytest

from gpu_common import bits, golden, inputs, same_bits_nan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    return m


def _both(cga, tag, strict, layout=None):
    s, g = inputs(tag), golden(tag)
    n, K = s["n"], int(g["K_full"])
    if layout is None:
        A = cga.Matrix.from_ell(n, K, s["diag"], g["ell_full_vals"], g["ell_full_cols"])
    else:
        A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], layout)
    b = cga.DeviceVector.from_host(s["source"])
    max_iter, tol = int(g["max_iter"]), float(g["tol"])
    # the new entry point first, on a matrix no SpMV has touched yet
    x2 = cga.DeviceVector.from_host(np.full(n, 7.0))
    hist_d = cga.DeviceVector.from_host(np.full(4 * max_iter, -3.0))
    status = cga.DeviceVector.from_host(np.full(4, -1, dtype=np.int32))
    cga.check(cga.lib().cgh_cg_solve_noalloc(A.h, b.ptr, x2.ptr, max_iter, tol, strict, hist_d.ptr, status.ptr), "cgh_cg_solve_noalloc")
    cga.check(cga.lib().cgh_sync())
    done = int(status.to_host()[0])
    h2 = hist_d.to_host().reshape(max_iter, 4)
    x1 = cga.DeviceVector(n)
    rep, hist = cga.cg_solve(A, b, x1, max_iter=max_iter, tol=tol, strict=strict)
    assert rep.ret > 0 and done == rep.iterations and done + 1 == rep.ret
    assert same_bits_nan(h2[:done + 1], hist)
    assert np.array_equal(bits(x2.to_host()), bits(x1.to_host()))
    return rep.ret, int(g["cg_ret"][4])


def test_two_calls_in_a_row_second_smaller(cga):
    ret, want = _both(cga, "cylinder_8000", strict=0)
    assert abs(ret - want) <= 1
    ret, want = _both(cga, "cylinder_2000", strict=0)     # smaller: the workspace still holds the 8000-row solve
    assert abs(ret - want) <= 1
    ret, want = _both(cga, "cylinder_2000", strict=1)     # strict sums of a short vector need no workspace of their own:
    assert ret == want                                    # the reference's trajectory bit for bit (golden, via cgh_cg_solve)
    ret, want = _both(cga, "poisson7_16", strict=0, layout=1)   # CSR layout through the same entry point


def test_refuses_what_it_cannot_hold(cga):
    import ctypes as C
    L = cga.lib()
    n = 128 * 128 * 80                                   # 1.3M rows > 2^20
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(128, 128, 80, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    hist = cga.DeviceVector(4 * 10); status = cga.DeviceVector(4, np.int32)
    import os
    os.environ["CG_HIP_SILENT_ERRORS"] = "1"
    try:
        assert L.cgh_cg_solve_noalloc(A.h, b.ptr, x.ptr, 10, 1e-6, 0, hist.ptr, status.ptr) != 0
        assert b"exceed" in L.cgh_last_error()
        s, g = inputs("cylinder_8000"), golden("cylinder_8000")
        A2 = cga.Matrix.from_ell(s["n"], int(g["K_full"]), s["diag"], g["ell_full_vals"], g["ell_full_cols"])
        b2 = cga.DeviceVector.from_host(s["source"]); x2 = cga.DeviceVector(s["n"])
        assert L.cgh_cg_solve_noalloc(A2.h, b2.ptr, x2.ptr, 10, 1e-6, 1, hist.ptr, status.ptr) != 0   # strict, 8000 rows >= 4096
        assert b"strict" in L.cgh_last_error()
        assert L.cgh_cg_solve_noalloc(A2.h, b2.ptr, x2.ptr, 0, 1e-6, 0, hist.ptr, status.ptr) != 0
    finally:
        del os.environ["CG_HIP_SILENT_ERRORS"]
