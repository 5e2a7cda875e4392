"""Order-exact parallel sums on the device (csrc/xsum.hip) against the reference's left-to-right vec_dot / vec_l1norm
(/root/reference/src/vec.c:67-85, restated in oracle/cg_oracle.c): bit-identical on every input, and the strict CG
trajectory built on them bit-identical to the reference loop at BASELINE's full size (256^3)."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_common import setknob, bits, golden, inputs, same_bits_nan
from test_xsum_model import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    return m


def seq_sum(t):
    one = np.ones_like(t)
    return O.lib().orc_vec_dot(O.D(np.ascontiguousarray(t)), O.D(one), t.size)


def eq(a, b):
    return (np.isnan(a) and np.isnan(b)) or np.float64(a).view(np.uint64) == np.float64(b).view(np.uint64)


@pytest.mark.parametrize("name,t", [c for c in cases() if c[1].size > 0], ids=[c[0] for c in cases() if c[1].size > 0])
def test_device_sum_equals_sequential(cga, monkeypatch, name, t):
    """sum t[i] as dot(t, 1) and, for non-negative data, as l1norm(t): forced through the parallel path at every size"""
    setknob(monkeypatch, "CGH_XSUM_MIN_ROWS", "0")
    t = np.ascontiguousarray(t, dtype=np.float64)
    n = t.size
    dt = cga.DeviceVector.from_host(t); d1 = cga.DeviceVector.from_host(np.ones(n))
    v = np.zeros(1)
    cga.binding.check(cga.lib().cgh_dot(dt.ptr, d1.ptr, n, 1, O.D(v)))
    assert eq(v[0], seq_sum(t)), (name, v[0], seq_sum(t))
    cga.binding.check(cga.lib().cgh_l1norm(dt.ptr, n, 1, O.D(v)))
    want = O.lib().orc_vec_l1norm(O.D(t), n)
    assert eq(v[0], want), (name, v[0], want)


def test_device_dot_products(cga, monkeypatch):
    rng = np.random.default_rng(5)
    for n in (70000, 1 << 20, (1 << 22) + 77):
        a = rng.standard_normal(n); b = a * (1.0 + 0.1 * rng.standard_normal(n))
        da = cga.DeviceVector.from_host(a); db = cga.DeviceVector.from_host(b)
        v = np.zeros(1)
        cga.binding.check(cga.lib().cgh_dot(da.ptr, db.ptr, n, 1, O.D(v)))
        assert eq(v[0], O.lib().orc_vec_dot(O.D(a), O.D(b), n))
        cga.binding.check(cga.lib().cgh_dot(da.ptr, da.ptr, n, 1, O.D(v)))
        assert eq(v[0], O.lib().orc_vec_dot(O.D(a), O.D(a), n))
        cga.binding.check(cga.lib().cgh_l1norm(db.ptr, n, 1, O.D(v)))
        assert eq(v[0], O.lib().orc_vec_l1norm(O.D(b), n))


@pytest.mark.parametrize("tag", ["cylinder_2000", "cylinder_32k", "poisson27_12x10x8"])
def test_strict_cg_through_parallel_sums(cga, monkeypatch, tag):
    """the golden trajectories again, with every strict sum forced through xsum.hip instead of the one-lane kernel"""
    setknob(monkeypatch, "CGH_XSUM_MIN_ROWS", "0")
    s, g = inputs(tag), golden(tag)
    n, K = s["n"], int(g["K_full"])
    A = cga.Matrix.from_ell(n, K, s["diag"], g["ell_full_vals"], g["ell_full_cols"])
    b = cga.DeviceVector.from_host(s["source"]); x = cga.DeviceVector(n)
    rep, hist = cga.cg_solve(A, b, x, max_iter=int(g["max_iter"]), tol=float(g["tol"]), strict=1, check_every=64)
    assert rep.ret == int(g["cg_ret"][4])
    assert same_bits_nan(hist, g["traj"])
    assert np.array_equal(bits(x.to_host()), bits(g["cg_x"]))


def test_full_size_27point_256cubed_strict_bit_exact(cga):
    """The wide-row kernels at full size against the reference's own code: 256^3 27-point (K = 26: tile-blocked copies of
    values and code words, x gathered from windows staged in LDS), one SpMV of a vector with entries of every magnitude
    and 8 strict CG iterations -- y, (res, rr, alpha, beta) and x bit-identical to oracle/_ref/libref.so (else the port)."""
    import ctypes as C
    import os
    N, iters = 256, 8
    n = N ** 3
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(N, N, N, 27, 0x5EED0001, cga.LAYOUT_ELL, b)
    ev, ec = A.download_ell(); diag = A.download_diag(); bh = b.to_host()
    K = A.K
    assert K == 26
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_so = os.path.join(root, "oracle", "_ref", "libref.so")
    if os.path.exists(ref_so):
        R = C.CDLL(ref_so)
        R.vec_dot.restype = C.c_double; R.vec_l1norm.restype = C.c_double
        R.vec_axpy.argtypes = [O.dp, O.dp, C.c_double, O.dp, C.c_int]
        mv = lambda v, out: R.mv_ell_symmetric_full_colmajor(n, K, O.D(diag), O.D(ev), O.I(ec), O.D(v), O.D(out))
        dot = lambda u, v: R.vec_dot(O.D(u), O.D(v), n)
        l1 = lambda u: R.vec_l1norm(O.D(u), n)
        axpy = lambda a_, b_, al, out: R.vec_axpy(O.D(a_), O.D(b_), al, O.D(out), n)
    else:
        L = O.lib()
        mv = lambda v, out: L.orc_mv_ell_symmetric_full_colmajor(n, K, O.D(diag), O.D(ev), O.I(ec), O.D(v), O.D(out))
        dot = lambda u, v: L.orc_vec_dot(O.D(u), O.D(v), n)
        l1 = lambda u: L.orc_vec_l1norm(O.D(u), n)
        axpy = lambda a_, b_, al, out: L.orc_vec_axpy(O.D(a_), O.D(b_), al, O.D(out), n)
    # one SpMV
    rng = np.random.default_rng(27)
    vh = rng.standard_normal(n) * np.exp2(rng.integers(-30, 30, n))
    want_y = np.empty(n)
    mv(vh, want_y)
    v = cga.DeviceVector.from_host(vh); y = cga.DeviceVector(n)
    A.spmv(v, y)
    assert A.dict_mode == 3 and A.stage_doubles > 0          # the kernels this test is about
    assert np.array_equal(bits(y.to_host()), bits(want_y))
    del v, y, vh, want_y
    # strict CG
    xr = np.zeros(n); r = bh.copy(); p = bh.copy(); Ap = np.empty(n)
    rr = dot(r, r)
    want = [[l1(r) / 2.0, rr, np.nan, np.nan]]
    for it in range(1, iters + 1):
        mv(p, Ap)
        alpha = rr / dot(p, Ap)
        axpy(xr, p, alpha, xr); axpy(r, Ap, -alpha, r)
        rr_new = dot(r, r)
        beta = rr_new / rr
        want.append([l1(r) / 2.0, rr_new, alpha, beta])
        axpy(r, p, beta, p)
        rr = rr_new
    rep, hist = cga.cg_solve(A, b, x, fixed_iters=iters, strict=1)
    assert same_bits_nan(hist[:iters + 1], np.array(want))
    assert np.array_equal(bits(x.to_host()), bits(xr))


def test_full_size_256cubed_strict_bit_exact(cga, monkeypatch):
    """BASELINE config 3 at full size: strict mode (parallel order-exact sums) against the reference's own arithmetic
    on the host (oracle/_ref/libref.so when present, else the oracle port), 50 iterations: (res, rr, alpha, beta) and x
    bit-identical.  Then a solve that CONVERGES at this size (tolerance taken from the host trajectory, iterations queued
    past it): the launch of the sums of r carries x += alpha p in workgroups nobody waits for, and at 256^3 the grid (2562
    workgroups) exceeds what the chip holds at once -- x at the converging iteration must equal the reference's all the
    same, with the x update riding along and without."""
    import ctypes as C
    import os
    N, iters = 256, 50
    n = N ** 3
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(N, N, N, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    ev, ec = A.download_ell(); diag = A.download_diag(); bh = b.to_host()
    K = A.K
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_so = os.path.join(root, "oracle", "_ref", "libref.so")
    if os.path.exists(ref_so):
        R = C.CDLL(ref_so)
        R.vec_dot.restype = C.c_double; R.vec_l1norm.restype = C.c_double
        R.vec_axpy.argtypes = [O.dp, O.dp, C.c_double, O.dp, C.c_int]
        mv = lambda v, out: R.mv_ell_symmetric_full_colmajor(n, K, O.D(diag), O.D(ev), O.I(ec), O.D(v), O.D(out))
        dot = lambda u, v: R.vec_dot(O.D(u), O.D(v), n)
        l1 = lambda u: R.vec_l1norm(O.D(u), n)
        axpy = lambda a_, b_, al, out: R.vec_axpy(O.D(a_), O.D(b_), al, O.D(out), n)
    else:
        L = O.lib()
        mv = lambda v, out: L.orc_mv_ell_symmetric_full_colmajor(n, K, O.D(diag), O.D(ev), O.I(ec), O.D(v), O.D(out))
        dot = lambda u, v: L.orc_vec_dot(O.D(u), O.D(v), n)
        l1 = lambda u: L.orc_vec_l1norm(O.D(u), n)
        axpy = lambda a_, b_, al, out: L.orc_vec_axpy(O.D(a_), O.D(b_), al, O.D(out), n)
    xr = np.zeros(n); r = bh.copy(); p = bh.copy(); Ap = np.empty(n)
    rr = dot(r, r)
    want = [[l1(r) / 2.0, rr, np.nan, np.nan]]
    low, stop_at, x_stop = want[0][0], 0, None     # first iteration >= 20 whose residual is below every earlier one
    for it in range(1, iters + 1):
        mv(p, Ap)
        alpha = rr / dot(p, Ap)
        axpy(xr, p, alpha, xr); axpy(r, Ap, -alpha, r)
        rr_new = dot(r, r)
        beta = rr_new / rr
        want.append([l1(r) / 2.0, rr_new, alpha, beta])
        if want[-1][0] < low:
            low = want[-1][0]
            if it >= 20 and stop_at == 0:
                stop_at, x_stop = it, xr.copy()
        axpy(r, p, beta, p)
        rr = rr_new
    rep, hist = cga.cg_solve(A, b, x, fixed_iters=iters, strict=1)
    print("\n[256^3 strict] %d iterations in %.2f ms = %.3f ms/iteration" % (iters, rep.solve_ms, rep.solve_ms / iters))
    assert same_bits_nan(hist, np.array(want))
    assert np.array_equal(bits(x.to_host()), bits(xr))
    assert stop_at > 0, "the residual of the 256^3 system did not fall between iterations 20 and 50"
    tol = float(np.nextafter(want[stop_at][0], np.inf))     # res[stop_at] < tol <= every earlier residual
    for xupd in ("1", "0"):
        setknob(monkeypatch, "CGH_XSUM_XUPD", xupd)
        x.zero()
        rep, hist = cga.cg_solve(A, b, x, max_iter=iters + 1, tol=tol, strict=1, check_every=32)
        assert rep.ret == stop_at + 1, (xupd, rep.ret, stop_at)
        ref = np.array(want[:stop_at + 1]); ref[stop_at, 3] = np.nan     # the converging iteration computes no beta
        assert same_bits_nan(hist, ref)
        assert np.array_equal(bits(x.to_host()), bits(x_stop)), "x at the converging iteration (CGH_XSUM_XUPD=%s)" % xupd
    setknob(monkeypatch, "CGH_XSUM_XUPD", "1")


def test_random_trials_device(cga, monkeypatch):
    """random lengths / scales / sign mixes, both the flat (few cells) and the grouped driver"""
    setknob(monkeypatch, "CGH_XSUM_MIN_ROWS", "0")
    rng = np.random.default_rng(2024)
    for trial in range(120):
        n = int(rng.integers(1, 60000)) if trial % 6 else int(rng.integers(530000, 1200000))
        kind = trial % 4
        if kind == 0:
            t = rng.standard_normal(n) ** 2 * 2.0 ** rng.integers(-40, 40)
        elif kind == 1:
            t = rng.standard_normal(n) * 2.0 ** rng.integers(-8, 8, n)
        elif kind == 2:
            t = np.round(rng.standard_normal(n) * 1024) / 1024      # many exactly representable ties
        else:
            t = rng.standard_normal(n) + 1.0
        dt = cga.DeviceVector.from_host(t); d1 = cga.DeviceVector.from_host(np.ones(n))
        v = np.zeros(1)
        cga.binding.check(cga.lib().cgh_dot(dt.ptr, d1.ptr, n, 1, O.D(v)))
        assert eq(v[0], seq_sum(t)), (trial, n, kind)
        cga.binding.check(cga.lib().cgh_dot(dt.ptr, dt.ptr, n, 1, O.D(v)))
        assert eq(v[0], O.lib().orc_vec_dot(O.D(t), O.D(t), n)), (trial, n, kind)
        cga.binding.check(cga.lib().cgh_l1norm(dt.ptr, n, 1, O.D(v)))
        assert eq(v[0], O.lib().orc_vec_l1norm(O.D(t), n)), (trial, n, kind)


def _hard_cases(rng):
    """adversarial inputs at the sizes that take the one-launch path (xs_fused: summaries and walk at once; > 2048 cells)"""
    n = 300000 + 77
    yield "cancellation", np.tile([1e16, 1.0, -1e16, 3.0], n // 4 + 1)[:n] * (1.0 + 1e-3 * rng.random(n))
    t = np.full(n, 2.0 ** -30); t[::4097] = -(2.0 ** -18)
    yield "sawtooth around a power of two", np.concatenate([[1.0 - 2.0 ** -20], t])
    yield "all ties", np.round(rng.standard_normal(700001) * 1024) / 1024
    t = rng.random(900000) * 2.0 ** -60; t[450000] = 1.0; t[450001:] *= 2.0 ** 58
    yield "tiny, then one large element, then comparable ones", t
    t = rng.random(400000); t[:270000] = 0.0
    yield "zero prefix over many groups", t
    t = rng.random(400000); t[123457] = np.inf
    yield "inf in the middle", t
    t = rng.random(400000); t[399999] = np.nan
    yield "nan at the end", t
    t = rng.standard_normal(64 * 64 * 128 * 3 + 1)
    yield "three full batches and one element", t
    yield "exactly two batches", rng.random(64 * 64 * 128 * 2) + 0.5
    yield "subnormal start", np.concatenate([np.full(300000, 5e-324), rng.random(100000) * 1e-300])


def test_fused_launch_hard_inputs_and_unfused_twin(cga, monkeypatch):
    """the one-launch path against the sequential sum, and against the two-launch path (CGH_XSUM_FUSED=0) bit for bit"""
    rng = np.random.default_rng(77)
    for name, t in _hard_cases(rng):
        t = np.ascontiguousarray(t, dtype=np.float64)
        n = t.size
        dt = cga.DeviceVector.from_host(t); d1 = cga.DeviceVector.from_host(np.ones(n))
        got = {}
        for fused in ("1", "0"):
            setknob(monkeypatch, "CGH_XSUM_FUSED", fused)
            v = np.zeros(3)
            cga.binding.check(cga.lib().cgh_dot(dt.ptr, d1.ptr, n, 1, O.D(v[0:1])))
            cga.binding.check(cga.lib().cgh_dot(dt.ptr, dt.ptr, n, 1, O.D(v[1:2])))
            cga.binding.check(cga.lib().cgh_l1norm(dt.ptr, n, 1, O.D(v[2:3])))
            got[fused] = v
        want = (seq_sum(t), O.lib().orc_vec_dot(O.D(t), O.D(t), n), O.lib().orc_vec_l1norm(O.D(t), n))
        for k in range(3):
            assert eq(got["1"][k], want[k]), (name, k, got["1"][k], want[k])
            assert eq(got["0"][k], want[k]), (name, k, "two launches")
    setknob(monkeypatch, "CGH_XSUM_FUSED", "1")


def test_fused_launch_more_than_64_batches(cga):
    """34M elements: 65 batches of 64 groups (the driver takes at most 64 at a time) and the group-prefix scan kernel"""
    rng = np.random.default_rng(78)
    n = 64 * 64 * 64 * 128 + 500001
    t = rng.random(n) + 0.25
    dt = cga.DeviceVector.from_host(t)
    v = np.zeros(1)
    cga.binding.check(cga.lib().cgh_dot(dt.ptr, dt.ptr, n, 1, O.D(v)))
    assert eq(v[0], O.lib().orc_vec_dot(O.D(t), O.D(t), n))
    cga.binding.check(cga.lib().cgh_l1norm(dt.ptr, n, 1, O.D(v)))
    assert eq(v[0], O.lib().orc_vec_l1norm(O.D(t), n))


def test_strict_cg_same_bits_with_and_without_the_fused_launches(cga, monkeypatch):
    """a strict solve that converges, above the group-level threshold (307 200 rows, odd variant too): sums in one launch
    with x += alpha p riding along (default), without the x update, and as two launches -- history, return value and x
    bit-identical (the converging iteration still gets its x update, later iterations none)"""
    for shape in ((80, 80, 48), (81, 79, 49)):
        n = shape[0] * shape[1] * shape[2]
        got = []
        for knobs in ({"CGH_XSUM_FUSED": "1", "CGH_XSUM_XUPD": "1"}, {"CGH_XSUM_FUSED": "1", "CGH_XSUM_XUPD": "0"},
                      {"CGH_XSUM_FUSED": "0", "CGH_XSUM_XUPD": "1"}):
            for k, v in knobs.items():
                setknob(monkeypatch, k, v)
            b = cga.DeviceVector(n); x = cga.DeviceVector(n)
            A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
            rep, hist = cga.cg_solve(A, b, x, max_iter=400, tol=1e-3, strict=1, check_every=16)
            got.append((rep.ret, hist.copy(), x.to_host()))
        assert 1 < got[0][0] < 400                        # converged well before max_iter: iterations were queued past it
        for other in got[1:]:
            assert other[0] == got[0][0]
            assert same_bits_nan(other[1], got[0][1])
            assert np.array_equal(bits(other[2]), bits(got[0][2]))
    setknob(monkeypatch, "CGH_XSUM_FUSED", "1")
    setknob(monkeypatch, "CGH_XSUM_XUPD", "1")


@pytest.mark.parametrize("shape", [(17, 17, 17), (20, 20, 11), (32, 32, 8), (64, 64, 33)])
def test_strict_cg_shapes_around_the_fused_paths(cga, shape):
    """default thresholds; n odd (scalar SpMV path, odd tail in the cell sums), n even but not a multiple of 128 (no
    per-cell sums from the SpMV), n a multiple of 128 (everything fused), and a size with the group level: the
    trajectory and x equal the oracle's bit for bit"""
    nx, ny, nz = shape
    n = nx * ny * nz
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(nx, ny, nz, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    ev, ec = A.download_ell(); diag = A.download_diag(); bh = b.to_host()
    ret, xo, traj = O.cg(O.FMT_ELL_FULL, n, diag, 0, ev, ec, None, A.K, bh, 400, 1e-7)
    rep, hist = cga.cg_solve(A, b, x, max_iter=400, tol=1e-7, strict=1)
    assert rep.ret == ret and ret > 0
    assert same_bits_nan(hist, traj) and np.array_equal(bits(x.to_host()), bits(xo))


def test_jacobi_graph_replay_matches_eager(cga, monkeypatch):
    """CGH_GRAPH=1 with the preconditioned kernels (device-side iteration counters)"""
    nx, ny, nz = 24, 20, 16
    n = nx * ny * nz
    b = cga.DeviceVector(n); x = cga.DeviceVector(n); x2 = cga.DeviceVector(n)
    A = cga.Matrix.poisson(nx, ny, nz, 27, 0x5EED0001, cga.LAYOUT_ELL, b)
    setknob(monkeypatch, "CGH_GRAPH", "0")
    rep, hist = cga.cg_solve(A, b, x, max_iter=300, tol=1e-8, strict=0, precond=1, check_every=16)
    setknob(monkeypatch, "CGH_GRAPH", "1")
    rep2, hist2 = cga.cg_solve(A, b, x2, max_iter=300, tol=1e-8, strict=0, precond=1, check_every=16)
    assert rep.ret > 0 and rep2.ret == rep.ret and same_bits_nan(hist2, hist)
    assert np.array_equal(bits(x2.to_host()), bits(x.to_host()))


@pytest.mark.parametrize("shape", [(16, 16, 16), (21, 15, 13), (32, 32, 31), (63, 65, 32), (64, 64, 64)])
def test_strict_cg_same_bits_with_the_vector_updates_in_the_sums_launches(cga, monkeypatch, shape):
    """Few cells (4096 .. 262 144 rows; odd lengths too): the launches of the sums keep their workgroups for the vector update
    behind the verdict (xs_tail: r -= alpha Ap behind p.Ap; x += alpha p and p = r + beta p behind the sums of r), so a strict
    iteration is three launches instead of five.  Against CGH_XSUM_TAIL=0 (cg_update_r / cg_update_xp as launches of their
    own): history, return value, x bit-identical -- converging solves (the converging iteration updates x and leaves p alone;
    iterations queued behind it do nothing) and a fixed number of iterations."""
    n = shape[0] * shape[1] * shape[2]
    for kw in (dict(max_iter=300, tol=1e-4, check_every=16), dict(fixed_iters=25)):
        got = []
        for tail in ("1", "0"):
            setknob(monkeypatch, "CGH_XSUM_TAIL", tail)
            b = cga.DeviceVector(n); x = cga.DeviceVector(n)
            A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
            rep, hist = cga.cg_solve(A, b, x, strict=1, **kw)
            got.append((rep.ret, hist.copy(), x.to_host()))
        assert got[0][0] == got[1][0], (shape, kw)
        assert same_bits_nan(got[0][1], got[1][1]), (shape, kw)
        assert np.array_equal(bits(got[0][2]), bits(got[1][2])), (shape, kw)
        if "tol" in kw:
            assert 1 < got[0][0] < 300
    setknob(monkeypatch, "CGH_XSUM_TAIL", "1")
