"""GPU parity of the device-resident layouts built by the device conversion (convert.hip): slot-major ELL,
full CSR (stream/adaptive kernel) and sliced ELL -- index arrays bit-exact vs the reference's conversion,
SpMV and strict-mode CG bit-exact -- plus size-independent checks at the BASELINE size (256^3)."""
import numpy as np
import pytest

import foamio
import oracle_lib as O
from gpu_common import setknob, GOLDENS, bits, golden, inputs, same_bits_nan

pytestmark = pytest.mark.gpu
LAYOUTS = [0, 1, 2]


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    return m


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("tag", GOLDENS)
def test_device_conversion_and_spmv(cga, tag, layout):
    s, g = inputs(tag), golden(tag)
    n = s["n"]
    A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], layout)
    assert (A.n, A.K, A.nnz_offdiag, A.layout) == (n, int(g["K_full"]), 2 * s["U"], layout)
    ev, ec = A.download_ell()
    assert np.array_equal(ec, g["ell_full_cols"]) and np.array_equal(bits(ev), bits(g["ell_full_vals"]))
    x = cga.DeviceVector.from_host(g["x_probe"]); y = cga.DeviceVector(n)
    A.spmv(x, y)
    assert np.array_equal(bits(y.to_host()), bits(g["y_probe"]))


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("tag", ["cylinder_2000", "cylinder_8000", "poisson7_16", "poisson27_12x10x8"])
def test_cg_strict_bit_exact_other_layouts(cga, tag, layout):
    s, g = inputs(tag), golden(tag)
    n = s["n"]
    A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], layout)
    b = cga.DeviceVector.from_host(s["source"]); x = cga.DeviceVector(n)
    rep, hist = cga.cg_solve(A, b, x, max_iter=int(g["max_iter"]), tol=float(g["tol"]), strict=1)
    assert rep.ret == int(g["cg_ret"][4]) and same_bits_nan(hist, g["traj"])
    assert np.array_equal(bits(x.to_host()), bits(g["cg_x"]))
    rep2, _ = cga.cg_solve(A, b, x, max_iter=int(g["max_iter"]), tol=float(g["tol"]), strict=0)
    assert abs(rep2.ret - rep.ret) <= 1


def _skewed(rng, n, maxlen):
    """upper COO with power-law-ish row lengths (a few rows far longer than the 64-entry wave threshold and the
    2048-entry window), unsorted"""
    want = np.minimum(maxlen, np.maximum(1, (1.0 / rng.random(n)).astype(np.int64)))
    want[rng.integers(0, n // 2)] = maxlen
    rows = np.repeat(np.arange(n), want)
    cols = rows + 1 + (rng.integers(0, 1 << 62, rows.size) % np.maximum(1, n - rows - 1))
    keep = cols < n
    pairs = np.unique(np.stack([rows[keep], cols[keep]], 1), axis=0)
    rng.shuffle(pairs)
    return pairs[:, 0].astype(np.int32).copy(), pairs[:, 1].astype(np.int32).copy()


@pytest.mark.parametrize("layout", [1, 2])
def test_skewed_rows(cga, layout, monkeypatch):
    rng = np.random.default_rng(99)
    n = 20000
    rows, cols = _skewed(rng, n, 6000)
    vals = -rng.random(rows.size); diag = 1.0 + rng.random(n) * 50; xh = rng.standard_normal(n)
    K = O.max_nnz_full(n, rows, cols)
    assert K > 4096                                   # exercises the block-per-row path of the CSR kernel
    want = O.mv_coo(n, diag, vals, rows, cols, xh)
    A = cga.Matrix.from_coo(n, diag, vals, rows, cols, layout)
    assert A.K == K and A.nnz_offdiag == 2 * rows.size
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    A.spmv(x, y)
    got = y.to_host()
    if layout == 2:
        assert np.array_equal(bits(got), bits(want))  # sliced ELL keeps the reference's per-row order
    else:
        # rows longer than 64 entries are summed by a wave / workgroup tree (documented in spmv_csr.hip):
        # exact for short rows, within a few ulp of the row's magnitude for long ones
        ev, ec, _ = O.coo_to_ell_full(n, vals, rows, cols, K)
        lens = (ec.reshape(K, n) >= 0).sum(0)
        short = lens <= 64
        assert np.array_equal(bits(got[short]), bits(want[short]))
        scale = np.abs(diag * xh) + np.abs(ev.reshape(K, n) * xh[np.maximum(ec.reshape(K, n), 0)]).sum(0)
        assert np.all(np.abs(got - want) <= 1e-14 * scale)


def test_poisson_256_conversion_matches_direct_generator(cga):
    """BASELINE size: the device COO -> ELL conversion of the 256^3 7-point matrix (50M upper entries) equals the
    directly generated ELL, and SpMV through ELL / CSR / sliced ELL agree bit for bit."""
    N, seed = 256, 0x5EED0001
    n = N ** 3
    b = cga.DeviceVector(n)
    A = cga.Matrix.poisson(N, N, N, 7, seed, cga.LAYOUT_ELL, b)
    y0 = cga.DeviceVector(n)
    A.spmv(b, y0)
    ref = y0.to_host()
    assert (A.K, A.nnz_offdiag) == (6, 100270080)
    for layout in (cga.LAYOUT_ELL, cga.LAYOUT_CSR, cga.LAYOUT_SELL):
        B = cga.Matrix.poisson(N, N, N, 7, seed, layout, None) if layout != cga.LAYOUT_ELL else None
        if B is None:
            continue
        assert (B.n, B.K, B.nnz_offdiag) == (n, 6, 100270080)
        y = cga.DeviceVector(n)
        B.spmv(b, y)
        assert np.array_equal(bits(y.to_host()), bits(ref)), layout
        if layout == cga.LAYOUT_CSR:
            ev, ec = B.download_ell()
            ev0, ec0 = A.download_ell()
            assert np.array_equal(ec, ec0) and np.array_equal(bits(ev), bits(ev0))
        B.free(); y.free()
    # linearity / symmetry properties at full size: x.(A y) == y.(A x) to rounding
    xs = cga.DeviceVector.from_host(foamio.hash01(7, np.arange(n)) - 0.5)
    Ax = cga.DeviceVector(n)
    A.spmv(xs, Ax)
    bh, xh = b.to_host(), xs.to_host()
    lhs, rhs = np.dot(xh, ref), np.dot(bh, Ax.to_host())
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("case", ["poisson7", "poisson27", "banded_random", "unstructured", "foam"])
def test_dictionary_coded_columns_same_bits(cga, monkeypatch, case):
    """ELL SpMV with 1-byte dictionary codes for the column ids (matrices with <= 255 distinct diagonals) gives the
    same bits as the int32 columns; matrices with more distinct offsets keep int32 columns."""
    rng = np.random.default_rng(8)
    if case == "poisson7":
        s = foamio.poisson_coo(20, 14, 10, 0x5EED0001, 7)
    elif case == "poisson27":
        s = foamio.poisson_coo(12, 10, 8, 0x5EED0001, 27)
    elif case == "foam":
        s = foamio.foam("8000")
    else:
        n = 6000
        offs = rng.choice(np.arange(1, 3000), 40, replace=False) if case == "banded_random" else None
        rr, cc = [], []
        for i in range(n - 1):
            if offs is not None:
                js = i + offs[rng.random(offs.size) < 0.2]
            else:
                js = rng.integers(i + 1, n, 3)
            js = np.unique(js[js < n])
            rr += [i] * js.size; cc += list(js)
        rows = np.array(rr, dtype=np.int32); cols = np.array(cc, dtype=np.int32)
        s = {"n": n, "U": rows.size, "rows": rows, "cols": cols, "upper": rng.standard_normal(rows.size),
             "diag": rng.standard_normal(n) + 5.0}
    n = s["n"]
    xh = rng.standard_normal(n)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    out = {}
    for flag in ("0", "1", "2"):      # int32 columns / one byte per entry / 4 slots of a row per 32-bit word
        setknob(monkeypatch, "CGH_ELL_DICT", flag)
        A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_ELL)
        assert A.dict_offsets == -1
        A.spmv(x, y)
        out[flag] = (y.to_host(), A.dict_offsets)
    want = O.mv_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], xh)
    for flag in ("0", "1", "2"):
        assert np.array_equal(bits(out[flag][0]), bits(want)), flag
    assert out["0"][1] == 0
    if case in ("poisson7", "poisson27", "banded_random"):
        assert 0 < out["1"][1] <= 255 and out["2"][1] == out["1"][1]
    elif case == "unstructured":
        assert out["1"][1] == 0 and out["2"][1] == 0


@pytest.mark.parametrize("generic", ["0", "1"])
@pytest.mark.parametrize("shape", [(12, 10, 8), (16, 16, 16), (64, 9, 3), (34, 6, 5)])
def test_tile_blocked_wide_rows_same_bits(cga, monkeypatch, shape, generic):
    """Wide banded rows (27-point: K = 26) read tile-blocked copies of the values and of the packed code words (dict_mode 3:
    the 26 slots of every 512-row tile together).  Same bits as the slot-major packed layout and as the reference's order,
    for row counts that are no multiple of the tile (last tile partly empty), with the compile-time and the run-time width;
    a strict solve (per-cell sums of p.Ap left by the SpMV where n is a multiple of 128) keeps its whole trajectory."""
    s = foamio.poisson_coo(*shape, 0x5EED0001, 27)
    n = s["n"]
    rng = np.random.default_rng(11)
    xh = rng.standard_normal(n)
    xh[rng.integers(0, n, 4)] = [np.inf, -np.inf, np.nan, -0.0]     # non-finite operands travel through LDS like any other
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    b = cga.DeviceVector.from_host(s["source"]); sol = cga.DeviceVector(n)
    setknob(monkeypatch, "CGH_ELL_GENERIC", generic)
    got = {}
    for blocked, stage, tiles in (("0", "1", "2"), ("1", "0", "2"), ("1", "1", "1"), ("1", "1", "3")):
        setknob(monkeypatch, "CGH_ELL_BLOCKED", blocked)
        setknob(monkeypatch, "CGH_ELL_STAGE", stage)
        setknob(monkeypatch, "CGH_ELL_STAGE_TILES", tiles)      # 3: two 512-row tiles per staging round whenever they fit
        A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_ELL)
        A.spmv(x, y)
        rep, hist = cga.cg_solve(A, b, sol, max_iter=40, tol=1e-9, strict=1)
        xs_ = sol.to_host()
        rep0, hist0 = cga.cg_solve(A, b, sol, max_iter=40, tol=1e-9, strict=0)
        got[blocked + stage + (tiles if blocked + stage == "11" else "")] = (A.dict_mode, y.to_host(), hist, xs_, hist0, A.stage_doubles)
    assert got["01"][0] == 2 and got["10"][0] == 3 and got["111"][0] == 3 and got["113"][0] == 3
    # gathers from LDS: the tile-blocked layout only (compile-time width 26, or the run-time-width kernel)
    assert got["01"][5] == 0 and got["10"][5] == 0 and got["111"][5] > 0 and got["113"][5] > got["111"][5]
    want = O.mv_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], xh)
    for k in got:
        assert same_bits_nan(got[k][1], want), k
        assert same_bits_nan(got["01"][2], got[k][2]), k      # strict: the whole trajectory, whatever kernel and launch shape
        # throughput mode: per-workgroup partials follow the launch shape (tiles per staging round): within rounding
        h0, hk = got["01"][4], got[k][4]
        m_ = min(len(h0), len(hk), 12)
        assert np.allclose(h0[:m_, :2], hk[:m_, :2], rtol=1e-11, atol=0), k
        assert np.array_equal(bits(got["01"][3]), bits(got[k][3])), k      # x of the strict solve


@pytest.mark.parametrize("K,n,spread", [(13, 3000, 40), (26, 2050, 900), (13, 1538, 8), (26, 5120, 5000), (9, 1100, 30), (18, 4096, 300),
                                        (31, 2600, 64)])
def test_general_banded_ell_of_compile_time_wide_widths(cga, monkeypatch, K, n, spread):
    """Any banded ELL matrix of a wide width (13, 26 slots: compile-time kernels; 9, 18, 31: the run-time-width ones; not symmetric, pads in the middle of rows, offsets far
    apart so that the staging plan has several windows or -- last case -- does not fit and the gathers stay with L1 / L2):
    every variant (slot-major packed codes, tile-blocked, tile-blocked + LDS-staged) adds each row's products in slot
    order, bit for bit as a plain loop does."""
    rng = np.random.default_rng(K * 1000 + n)
    offs = np.unique(np.concatenate([rng.integers(-spread, spread + 1, 3 * K), [-1, 1]]))
    offs = rng.permutation(offs[offs != 0])[:K]
    assert offs.size == K
    i = np.arange(n)
    ec = np.full((K, n), -1, dtype=np.int32)
    ev = np.zeros((K, n))
    for s_ in range(K):
        c = i + offs[s_]
        ok = (c >= 0) & (c < n) & (rng.random(n) < 0.9)      # pads wherever the column falls outside, and at random
        ec[s_, ok] = c[ok]
        ev[s_, ok] = rng.standard_normal(int(ok.sum()))
    diag = rng.standard_normal(n) + 4.0
    xh = rng.standard_normal(n)
    want = 0.0 + diag * xh
    for s_ in range(K):
        ok = ec[s_] >= 0
        t = want + ev[s_] * xh[np.where(ok, ec[s_], 0)]
        want = np.where(ok, t, want)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    staged = {}
    for blocked, stage in (("0", "0"), ("1", "0"), ("1", "1")):
        setknob(monkeypatch, "CGH_ELL_BLOCKED", blocked)
        setknob(monkeypatch, "CGH_ELL_STAGE", stage)
        A = cga.Matrix.from_ell(n, K, diag, ev.ravel(), ec.ravel())
        A.spmv(x, y)
        assert A.dict_mode == (3 if blocked == "1" else 2)
        assert np.array_equal(bits(y.to_host()), bits(want)), (blocked, stage)
        staged[blocked + stage] = A.stage_doubles
    assert staged["00"] == 0 and staged["10"] == 0
    assert (staged["11"] > 0) == (spread < 5000)


@pytest.mark.parametrize("shape,points", [((20, 14, 10), 7), ((64, 8, 4), 7), ((16, 16, 16), 7), ((130, 3, 2), 7)])
def test_narrow_rows_gathering_from_lds_same_bits(cga, monkeypatch, shape, points):
    """CGH_ELL_STAGE=2: the kernel that stages x in LDS also serves narrow byte-coded rows (off by default: no gain there).
    Same bits as the gathers from L1 / L2 -- SpMV with non-finite entries in x, and a strict and a throughput-mode solve."""
    s = foamio.poisson_coo(*shape, 0x5EED0001, points)
    n = s["n"]
    rng = np.random.default_rng(12)
    xh = rng.standard_normal(n)
    xh[rng.integers(0, n, 3)] = [np.inf, -np.inf, np.nan]
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    b = cga.DeviceVector.from_host(s["source"]); sol = cga.DeviceVector(n)
    got = {}
    for stage in ("1", "2"):
        setknob(monkeypatch, "CGH_ELL_STAGE", stage)
        A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_ELL)
        A.spmv(x, y)
        rep, hist = cga.cg_solve(A, b, sol, max_iter=30, tol=1e-9, strict=1)
        xs = sol.to_host()
        rep0, hist0 = cga.cg_solve(A, b, sol, max_iter=30, tol=1e-9, strict=0)
        got[stage] = (A.stage_doubles, y.to_host(), hist, xs, hist0)
    assert got["1"][0] == 0 and got["2"][0] >= 512
    assert same_bits_nan(got["1"][1], got["2"][1])
    assert same_bits_nan(got["1"][2], got["2"][2]) and same_bits_nan(got["1"][4], got["2"][4])
    assert np.array_equal(bits(got["1"][3]), bits(got["2"][3]))


def test_two_solvers_interleaved_reproduce_their_solo_bits(cga):
    """Nothing about a launch travels through process-wide state (cgh_spmv_opts): a strict and a throughput-mode solver on
    the SAME matrix, their iteration batches interleaved, give exactly the histories and solutions each gives alone
    (the strict one bit-identical to the reference's golden trajectory).  The strict one leaves per-cell sums of p.Ap for
    its order-exact dot (n is a multiple of 128 and above the parallel-sum threshold), the other per-workgroup partials."""
    s, g = inputs("cylinder_8000"), golden("cylinder_8000")
    n, K = s["n"], int(g["K_full"])
    A = cga.Matrix.from_ell(n, K, s["diag"], g["ell_full_vals"], g["ell_full_cols"])
    b = cga.DeviceVector.from_host(s["source"])
    cap = int(g["max_iter"])
    solo = {}
    for strict in (1, 0):
        x = cga.DeviceVector(n)
        sv = cga.Solver(A, cap)
        sv.start(b, x, tol=float(g["tol"]), strict=strict)
        for _ in range(20):
            sv.iterate(30)
        done, queued = sv.poll()
        assert 0 < done < queued
        solo[strict] = (done, sv.history(0, done + 1), x.to_host())   # entries after the converging iteration are not written
        sv.free()
    xs = {1: cga.DeviceVector(n), 0: cga.DeviceVector(n)}
    svs = {1: cga.Solver(A, cap), 0: cga.Solver(A, cap)}
    for strict in (1, 0):
        svs[strict].start(b, xs[strict], tol=float(g["tol"]), strict=strict)
    for k in range(20):                                   # interleave: 7 + 23 iterations of one, 30 of the other
        svs[1].iterate(7); svs[0].iterate(30); svs[1].iterate(23)
    for strict in (1, 0):
        done, queued = svs[strict].poll()
        assert done == solo[strict][0] and done > 0
        assert same_bits_nan(svs[strict].history(0, done + 1), solo[strict][1])
        assert np.array_equal(bits(xs[strict].to_host()), bits(solo[strict][2]))
        svs[strict].free()
    assert solo[1][0] + 1 == int(g["cg_ret"][4]) and same_bits_nan(solo[1][1], g["traj"])
