"""Jacobi-preconditioned CG (cgh_cg_opts.precond).  The reference has no preconditioner, so there is
no reference output to pin this against -- PARITY UNPINNED by construction.  Checked instead against a plain numpy
PCG (same recurrence, same l1norm(r)/2 stop test; tolerance 1e-10 relative over the first iterations: only the
summation order differs), against a sparse direct solve, and for doing nothing when switched off."""
import os

import numpy as np
import pytest

import foamio
from gpu_common import bits, same_bits_nan
from test_mm_reader import MTX

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    return m


def numpy_pcg(A, d, b, max_iter, tol):
    n = b.size
    x = np.zeros(n); r = b.copy(); z = r / d; p = z.copy(); rz = r @ z
    hist = [[np.abs(r).sum() / 2.0, rz, np.nan, np.nan]]
    for it in range(1, max_iter):
        Ap = A @ p
        alpha = rz / (p @ Ap)
        x += alpha * p; r -= alpha * Ap
        res = np.abs(r).sum() / 2.0
        z = r / d; rz_new = r @ z
        if res < tol:
            hist.append([res, rz_new, alpha, np.nan])
            return it + 1, x, np.array(hist)
        beta = rz_new / rz
        hist.append([res, rz_new, alpha, beta])
        p = z + beta * p; rz = rz_new
    return -1, x, np.array(hist)


def systems():
    sp = pytest.importorskip("scipy.sparse")
    for name in ("plate", "poisson"):
        s = foamio.mtx_split(os.path.join(MTX, "plate_24x18.mtx")) if name == "plate" else foamio.poisson_coo(14, 12, 10, 0x5EED0001, 7)
        n = s["n"]
        up = sp.coo_matrix((s["upper"], (s["rows"], s["cols"])), shape=(n, n))
        A = (sp.diags(s["diag"]) + up + up.T).tocsr()
        b = s["source"] if "source" in s else np.random.default_rng(3).random(n) - 0.5
        yield name, s, A, b


@pytest.mark.parametrize("layout", ["LAYOUT_ELL", "LAYOUT_CSR"])
def test_jacobi_pcg_matches_numpy_and_direct_solve(cga, layout):
    spl = pytest.importorskip("scipy.sparse.linalg")
    for name, s, A, b in systems():
        n = s["n"]
        M = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], getattr(cga, layout))
        db = cga.DeviceVector.from_host(b); dx = cga.DeviceVector(n)
        rep, hist = cga.cg_solve(M, db, dx, max_iter=2000, tol=1e-9, strict=0, precond=1)
        ret, xn, hn = numpy_pcg(A, s["diag"], b, 2000, 1e-9)
        assert rep.ret > 0 and abs(rep.ret - ret) <= 1, (name, rep.ret, ret)
        m = min(12, hist.shape[0], hn.shape[0])
        rel = np.abs(hist[:m, :2] - hn[:m, :2]) / np.abs(hn[:m, :2])
        assert rel.max() < 1e-10, (name, rel.max())
        xd = spl.spsolve(A.tocsc(), b)
        assert np.abs(dx.to_host() - xd).max() <= 1e-7 * np.abs(xd).max(), name
        # the point of the hook: fewer iterations than the reference's unpreconditioned loop on a badly scaled matrix
        rep0, _ = cga.cg_solve(M, db, dx, max_iter=2000, tol=1e-9, strict=0)
        print("\n[%s %s] iterations: jacobi %d, none %d" % (name, layout, rep.ret, rep0.ret))
        if name == "plate":
            assert rep0.ret == -1 or rep.ret < rep0.ret


def test_precond_off_is_the_reference_loop_and_strict_is_refused(cga):
    s = foamio.poisson_coo(14, 12, 10, 0x5EED0001, 7)
    n = s["n"]
    M = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_ELL)
    b = cga.DeviceVector.from_host(s["source"]); x = cga.DeviceVector(n); x2 = cga.DeviceVector(n)
    rep, hist = cga.cg_solve(M, b, x, max_iter=500, tol=1e-8, strict=1)
    rep2, hist2 = cga.cg_solve(M, b, x2, max_iter=500, tol=1e-8, strict=1, precond=0)
    assert rep2.ret == rep.ret and same_bits_nan(hist, hist2) and np.array_equal(bits(x.to_host()), bits(x2.to_host()))
    with pytest.raises(RuntimeError):
        cga.cg_solve(M, b, x, max_iter=10, tol=1e-8, strict=1, precond=1)
    with pytest.raises(RuntimeError):
        cga.cg_solve(M, b, x, max_iter=10, tol=1e-8, strict=0, precond=7)
