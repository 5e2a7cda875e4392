#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (oracle/_ref/libref.so,
compiled by oracle/Makefile from /root/reference/src).  Runs only in the build
container; the outputs are data (inputs + expected outputs) and are committed.

For every input system it records, from the reference's own functions:
  K_full / K_upper                      compute_max_nnz_row_{full,upper}
  ell_full_{vals,cols}                  coo_to_ell_symmetric_full_colmajor
  ell_upcm_{vals,cols}                  coo_to_ell_symmetric_upper_colmajor
  ell_uprm_{vals,cols}                  coo_to_ell_symmetric_upper
  row_ptr                               coo_to_csr
  x_probe / y_probe                     mv_ell_symmetric_full_colmajor on a stored x
                                        (and checks COO/CSR/upper-ELL SpMV give the same bits)
  cg_ret[fmt], cg_x                     conjugate_gradient_{coo,csr,ell,ell_colmajor,ell_full_colmajor}
  traj                                  full-precision per-iteration (l1/2, r.r, alpha, beta),
                                        from the same loop as src/cg.c:414-508 driven through the
                                        reference's mv_*/vec_* (cg.c itself only prints %g); the final x
                                        of that loop is asserted bit-equal to the solver's x.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import foamio  # noqa: E402

ref = C.CDLL(os.path.join(HERE, "_ref", "libref.so"))
libc = C.CDLL(None)

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def D(a):
    return a.ctypes.data_as(dp)


def I(a):
    return a.ctypes.data_as(ip)


ref.vec_dot.restype = C.c_double
ref.vec_l1norm.restype = C.c_double
ref.vec_axpy.argtypes = [dp, dp, C.c_double, dp, C.c_int]
for name in ("conjugate_gradient_coo", "conjugate_gradient_csr"):
    getattr(ref, name).argtypes = [C.c_int, dp, C.c_int, dp, ip, ip, dp, dp, C.c_int, C.c_double]
for name in ("conjugate_gradient_ell", "conjugate_gradient_ell_colmajor",
             "conjugate_gradient_ell_full_colmajor"):
    getattr(ref, name).argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]


class quiet:
    """silence the reference's printf chatter (C stdio, fd 1)"""

    def __enter__(self):
        libc.fflush(None)
        self.saved = os.dup(1)
        null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(null, 1)
        os.close(null)

    def __exit__(self, *a):
        libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def cg_traj_full_ell(n, diag, ev, ec, K, b, max_iter, tol):
    """src/cg.c:414-508 step for step, through the reference's own kernels."""
    x = np.zeros(n)
    r = np.empty(n)
    p = np.empty(n)
    Ap = np.empty(n)
    ref.mv_ell_symmetric_full_colmajor(n, K, D(diag), D(ev), I(ec), D(x), D(r))
    ref.vec_sub(D(b), D(r), D(r), n)
    ref.vec_assign(D(p), D(r), n)
    rr = ref.vec_dot(D(r), D(r), n)
    traj = [[ref.vec_l1norm(D(r), n) / 2.0, rr, np.nan, np.nan]]
    ret = -1
    for it in range(1, max_iter):
        ref.mv_ell_symmetric_full_colmajor(n, K, D(diag), D(ev), I(ec), D(p), D(Ap))
        alpha = rr / ref.vec_dot(D(p), D(Ap), n)
        ref.vec_axpy(D(x), D(p), alpha, D(x), n)
        ref.vec_axpy(D(r), D(Ap), -alpha, D(r), n)
        rr_new = ref.vec_dot(D(r), D(r), n)
        res = ref.vec_l1norm(D(r), n) / 2.0
        if res < tol:
            traj.append([res, rr_new, alpha, np.nan])
            ret = it + 1
            break
        beta = rr_new / rr
        traj.append([res, rr_new, alpha, beta])
        ref.vec_axpy(D(r), D(p), beta, D(p), n)
        rr = rr_new
    return ret, x, np.array(traj)


def golden_for(sysd, tag, max_iter=1000, tol=1e-6, probe_seed=1234):
    n, U = sysd["n"], sysd["U"]
    diag, upper, rows, cols, b = (np.ascontiguousarray(sysd[k]) for k in
                                  ("diag", "upper", "rows", "cols", "source"))
    out = dict(n=n, U=U)
    with quiet():
        Kf = ref.compute_max_nnz_row_full(n, U, I(rows), I(cols))
        Ku = ref.compute_max_nnz_row_upper(n, U, I(rows), I(cols))
    out["K_full"], out["K_upper"] = Kf, Ku

    ev = np.empty(n * Kf); ec = np.empty(n * Kf, dtype=np.int32)
    uv = np.empty(n * Ku); uc = np.empty(n * Ku, dtype=np.int32)
    rv = np.empty(n * Ku); rc = np.empty(n * Ku, dtype=np.int32)
    rp = np.empty(n + 1, dtype=np.int32)
    with quiet():
        ref.coo_to_ell_symmetric_full_colmajor(n, U, D(upper), I(rows), I(cols), D(ev), I(ec), Kf)
        ref.coo_to_ell_symmetric_upper_colmajor(n, U, D(upper), I(rows), I(cols), D(uv), I(uc), Ku)
        ref.coo_to_ell_symmetric_upper(n, U, D(upper), I(rows), I(cols), D(rv), I(rc), Ku)
        ref.coo_to_csr(n, U, D(upper), I(rows), I(cols), I(rp))
    out.update(ell_full_vals=ev, ell_full_cols=ec, ell_upcm_vals=uv, ell_upcm_cols=uc,
               ell_uprm_vals=rv, ell_uprm_cols=rc, row_ptr=rp)

    xp = np.random.default_rng(probe_seed).standard_normal(n)
    ys = []
    for _ in range(5):
        ys.append(np.empty(n))
    with quiet():
        ref.mv_ell_symmetric_full_colmajor(n, Kf, D(diag), D(ev), I(ec), D(xp), D(ys[0]))
        ref.mv_coo_symmetric(n, U, D(diag), D(upper), I(rows), I(cols), D(xp), D(ys[1]))
        ref.mv_csr_symmetric(n, D(diag), D(upper), I(cols), I(rp), D(xp), D(ys[2]))
        ref.mv_ell_symmetric_upper_colmajor(n, Ku, D(diag), D(uv), I(uc), D(xp), D(ys[3]))
        ref.mv_ell_symmetric_upper(n, Ku, D(diag), D(rv), I(rc), D(xp), D(ys[4]))
    same = [bool(np.array_equal(ys[0].view(np.uint64), y.view(np.uint64))) for y in ys]
    out["x_probe"], out["y_probe"] = xp, ys[0]
    out["y_probe_coo"], out["y_probe_upcm"] = ys[1], ys[3]
    out["spmv_bits_equal_full_coo_csr_upcm_uprm"] = np.array(same)

    rets = []
    xs = []
    with quiet():
        for fmt in range(5):
            x = np.full(n, 7.0)  # the reference zeroes x itself (cg.c:433)
            if fmt == 0:
                r_ = ref.conjugate_gradient_coo(n, D(diag), U, D(upper), I(rows), I(cols), D(b), D(x), max_iter, tol)
            elif fmt == 1:
                r_ = ref.conjugate_gradient_csr(n, D(diag), U, D(upper), I(rp), I(cols), D(b), D(x), max_iter, tol)
            elif fmt == 2:
                r_ = ref.conjugate_gradient_ell(n, D(diag), U, D(rv), I(rc), Ku, D(b), D(x), max_iter, tol)
            elif fmt == 3:
                r_ = ref.conjugate_gradient_ell_colmajor(n, D(diag), U, D(uv), I(uc), Ku, D(b), D(x), max_iter, tol)
            else:
                r_ = ref.conjugate_gradient_ell_full_colmajor(n, D(diag), U, D(ev), I(ec), Kf, D(b), D(x), max_iter, tol)
            rets.append(r_)
            xs.append(x)
        ret, xt, traj = cg_traj_full_ell(n, diag, ev, ec, Kf, b, max_iter, tol)
    assert ret == rets[4], (ret, rets)
    assert np.array_equal(xt.view(np.uint64), xs[4].view(np.uint64)), "traj loop != reference solver"
    out["cg_ret"] = np.array(rets)
    out["cg_x"] = xs[4]
    out["cg_x_bits_equal_across_fmt"] = np.array(
        [bool(np.array_equal(xs[4].view(np.uint64), x.view(np.uint64))) for x in xs])
    out["traj"] = traj
    out["max_iter"], out["tol"] = max_iter, tol
    path = os.path.join(foamio.GOLDEN, "golden_%s.npz" % tag)
    np.savez_compressed(path, **out)
    print("%-14s n=%d U=%d Kf=%d Ku=%d ret=%s spmv_same=%s x_same=%s iters=%d -> %s" % (
        tag, n, U, Kf, Ku, rets, same, out["cg_x_bits_equal_across_fmt"].tolist(),
        traj.shape[0] - 1, os.path.basename(path)))


def capture_stdout(fn):
    """run fn() with C stdout redirected into a file; return the text"""
    import tempfile
    libc.fflush(None)
    saved = os.dup(1)
    with tempfile.TemporaryFile() as tf:
        os.dup2(tf.fileno(), 1)
        try:
            fn()
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tf.seek(0)
        return tf.read().decode()


def golden_stdout(sysd, tag, max_iter=1000, tol=1e-6):
    """the text the reference's solvers print (banner, r[0..4], per-iteration residuals, final line)"""
    n, U = sysd["n"], sysd["U"]
    diag, upper, rows, cols, b = (np.ascontiguousarray(sysd[k]) for k in ("diag", "upper", "rows", "cols", "source"))
    g = np.load(os.path.join(foamio.GOLDEN, "golden_%s.npz" % tag))
    Kf = int(g["K_full"])
    ev = np.ascontiguousarray(g["ell_full_vals"]); ec = np.ascontiguousarray(g["ell_full_cols"])
    rp = np.ascontiguousarray(g["row_ptr"])
    x = np.zeros(n)
    texts = {
        "ell_full": capture_stdout(lambda: ref.conjugate_gradient_ell_full_colmajor(n, D(diag), U, D(ev), I(ec), Kf, D(b), D(x), max_iter, tol)),
        "coo": capture_stdout(lambda: ref.conjugate_gradient_coo(n, D(diag), U, D(upper), I(rows), I(cols), D(b), D(x), max_iter, tol)),
        "csr": capture_stdout(lambda: ref.conjugate_gradient_csr(n, D(diag), U, D(upper), I(rp), I(cols), D(b), D(x), max_iter, tol)),
        "coo_to_csr": capture_stdout(lambda: ref.coo_to_csr(n, U, D(upper), I(rows), I(cols), I(rp.copy()))),
    }
    for k, t in texts.items():
        with open(os.path.join(foamio.GOLDEN, "stdout_%s_%s.txt" % (tag, k)), "w") as f:
            f.write(t)
        print("stdout %s/%s: %d lines" % (tag, k, t.count("\n")))


def main():
    for name in ("2000", "8000", "32k"):
        golden_for(foamio.foam(name), "cylinder_" + name)
    # synthetic inputs from this repo's seeded generators, solved BY THE REFERENCE
    golden_for(foamio.poisson_coo(16, 16, 16, 0x5EED0001, 7), "poisson7_16", max_iter=400, tol=1e-6)
    golden_for(foamio.poisson_coo(32, 24, 20, 0x5EED0001, 7), "poisson7_32x24x20", max_iter=400, tol=1e-6)
    golden_for(foamio.poisson_coo(12, 10, 8, 0x5EED0001, 27), "poisson27_12x10x8", max_iter=400, tol=1e-6)
    golden_stdout(foamio.foam("2000"), "cylinder_2000")
    golden_stdout(foamio.poisson_coo(12, 10, 8, 0x5EED0001, 27), "poisson27_12x10x8", max_iter=400)


if __name__ == "__main__":
    main()
