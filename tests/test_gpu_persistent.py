"""GPU parity of the resident launch of small systems (csrc/cg_persistent.h): throughput-mode CG of at most 131 072 rows
runs its iterations inside ONE launch per batch (workgroups keep their rows in registers and exchange partial sums and
the new r of off-tile columns as tagged words).  Bar: bit-identical -- whole history (res, r.r, alpha, beta), x, the
converging iteration and the return value -- to the three launches per iteration it replaces (cgh_config_persistent(0)),
on every golden system, on shapes around the tile boundaries, across batch boundaries and when the two paths take turns
inside one solve; and the goldens' trajectories to the tolerance of the throughput mode (reference loop: src/cg.c:414-508)."""
import numpy as np
import pytest

from gpu_common import GOLDENS, bits, golden, inputs, same_bits_nan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    return m


def _solve(cga, A, b_host, n, on, **kw):
    cga.check(cga.lib().cgh_config_persistent(on))
    b = cga.DeviceVector.from_host(b_host)
    x = cga.DeviceVector.from_host(np.full(n, 3.0))   # the solver must zero it (cg.c:433)
    before = cga.lib().cgh_persistent_launches()
    rep, hist = cga.cg_solve(A, b, x, strict=0, **kw)
    took = cga.lib().cgh_persistent_launches() - before
    # the path under test really is the one that ran (rows wider than 8 slots keep three launches either way)
    assert (took > 0) == bool(on and A.K <= 8), (on, A.K, took)
    return rep, hist, x.to_host()


@pytest.mark.parametrize("tag", GOLDENS)
def test_resident_launch_same_bits_as_three_launches_on_goldens(cga, tag):
    s, g = inputs(tag), golden(tag)
    n, K = s["n"], int(g["K_full"])
    A = cga.Matrix.from_ell(n, K, s["diag"], g["ell_full_vals"], g["ell_full_cols"])
    kw = dict(max_iter=int(g["max_iter"]), tol=float(g["tol"]))
    rep0, h0, x0 = _solve(cga, A, s["source"], n, 0, **kw)
    rep1, h1, x1 = _solve(cga, A, s["source"], n, 1, **kw)
    if K > 8:
        pytest.skip("27-point rows are wider than the resident launch keeps in registers: both runs took three launches")
    assert rep1.ret == rep0.ret and rep1.iterations == rep0.iterations
    assert same_bits_nan(h1, h0)
    assert np.array_equal(bits(x1), bits(x0))
    # and the reference's trajectory within the throughput mode's tolerance over the first iterations, same return value
    want_ret = int(g["cg_ret"][4])
    assert rep1.ret == want_ret or (want_ret > 0 and abs(rep1.ret - want_ret) <= 1)
    m = min(10, h1.shape[0], g["traj"].shape[0])
    assert np.allclose(h1[:m, 0], g["traj"][:m, 0], rtol=1e-10, atol=0.0)
    # looking more often (one launch per `check_every` iterations in the three-launch path) changes nothing
    rep2, h2, x2 = _solve(cga, A, s["source"], n, 1, check_every=1, **kw)
    assert rep2.ret == rep1.ret and same_bits_nan(h2, h1) and np.array_equal(bits(x2), bits(x1))


# shapes: one lane pair; less than a wave; one tile minus / exactly / plus one pair; a ragged last tile; 63, 250 and 256 tiles
SHAPES = [(2, 1, 1), (6, 5, 1), (17, 15, 2), (16, 16, 2), (257, 2, 1), (40, 20, 10), (32, 32, 31), (64, 64, 31), (64, 64, 32)]


@pytest.mark.parametrize("shape", SHAPES)
def test_resident_launch_same_bits_across_tile_shapes(cga, shape):
    nx, ny, nz = shape
    n = nx * ny * nz
    b = cga.DeviceVector(n)
    A = cga.Matrix.poisson(nx, ny, nz, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    bh = b.to_host()
    for kw in (dict(fixed_iters=40), dict(max_iter=400, tol=1e-9 * np.abs(bh).sum())):
        rep0, h0, x0 = _solve(cga, A, bh, n, 0, **kw)
        rep1, h1, x1 = _solve(cga, A, bh, n, 1, **kw)
        assert rep1.ret == rep0.ret and rep1.iterations == rep0.iterations, (shape, kw)
        assert same_bits_nan(h1, h0), (shape, kw)
        # (a system of a few rows is solved exactly long before 40 iterations are over; from then on alpha = 0/0 and
        # everything is NaN on both paths -- NaNs are compared as NaNs, their sign and payload are not part of the bar)
        assert same_bits_nan(x1, x0), (shape, kw)
    assert rep1.ret > 0          # the second case converges: x is advanced at the converging iteration, p is not
    assert np.all(np.isfinite(x1)) and np.array_equal(bits(x1), bits(x0))


def test_batches_and_the_two_paths_taking_turns(cga):
    """One solve queued as 1 + 7 + 32 iterations, each batch by the other path in turn, equals 40 iterations of either path:
    the resident launch hands x, r, p and the history over exactly as the three launches leave them (and takes them back)."""
    nx, ny, nz = 48, 40, 24
    n = nx * ny * nz
    b = cga.DeviceVector(n)
    A = cga.Matrix.poisson(nx, ny, nz, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    L = cga.lib()

    def run(plan):
        x = cga.DeviceVector(n)
        sv = cga.Solver(A, 64)
        sv.start(b, x, tol=-1.0, strict=0)
        for on, count in plan:
            cga.check(L.cgh_config_persistent(on))
            sv.iterate(count, 0)
        done, queued = sv.poll()
        assert done == 0 and queued == 40
        return sv.history(0, 41), x.to_host()

    h_ref, x_ref = run([(0, 40)])
    for plan in ([(1, 40)], [(1, 1), (1, 7), (1, 32)], [(1, 1), (0, 7), (1, 32)], [(0, 5), (1, 30), (0, 5)]):
        h, x = run(plan)
        assert np.array_equal(bits(h), bits(h_ref)), plan
        assert np.array_equal(bits(x), bits(x_ref)), plan


def test_frozen_after_convergence_and_restart(cga):
    """Iterations queued behind the converging one do nothing (the launch sees the flag and leaves at once), and a second
    solve on the same solver object starts clean (new epoch: none of the first solve's tagged words can be mistaken)."""
    nx, ny, nz = 32, 32, 8
    n = nx * ny * nz
    b = cga.DeviceVector(n)
    A = cga.Matrix.poisson(nx, ny, nz, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    cga.check(cga.lib().cgh_config_persistent(1))
    tol = 1e-8 * np.abs(b.to_host()).sum()
    x = cga.DeviceVector(n)
    sv = cga.Solver(A, 600)
    outs = []
    for _ in range(2):
        sv.start(b, x, tol=tol, strict=0)
        sv.iterate(200, 0)
        sv.iterate(200, 0)          # frozen
        done, queued = sv.poll()
        assert 0 < done < 200 and queued == 400
        outs.append((done, sv.history(0, done + 1), x.to_host()))
    assert outs[0][0] == outs[1][0]
    assert same_bits_nan(outs[0][1], outs[1][1]) and np.array_equal(bits(outs[0][2]), bits(outs[1][2]))
    # the residual really is below the tolerance, and the iteration after it was not run
    assert outs[0][1][-1, 0] < tol <= outs[0][1][-2, 0]


def test_unstructured_columns_int32(cga):
    """The OpenFOAM systems have no small set of column offsets (int32 columns, K = 4..6) and columns far from the diagonal:
    most gathers cross tiles.  cylinder_32k (63 tiles, ragged last tile), 300 fixed iterations."""
    s, g = inputs("cylinder_32k"), golden("cylinder_32k")
    n, K = s["n"], int(g["K_full"])
    A = cga.Matrix.from_ell(n, K, s["diag"], g["ell_full_vals"], g["ell_full_cols"])
    rep0, h0, x0 = _solve(cga, A, s["source"], n, 0, fixed_iters=300)
    rep1, h1, x1 = _solve(cga, A, s["source"], n, 1, fixed_iters=300)
    assert A.dict_offsets <= 0            # int32 columns
    assert same_bits_nan(h1, h0) and np.array_equal(bits(x1), bits(x0))


def test_history_window_slides_under_the_resident_launch(cga):
    """A solve longer than the device's history window (4096 iterations) continues in a fresh window: iteration numbers --
    and with them the tags of the exchanged words -- start again at 1 inside the same solve.  9000 iterations of both paths."""
    nx, ny, nz = 64, 64, 31
    n = nx * ny * nz
    b = cga.DeviceVector(n)
    A = cga.Matrix.poisson(nx, ny, nz, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    bh = b.to_host()
    rep0, h0, x0 = _solve(cga, A, bh, n, 0, fixed_iters=9000)
    rep1, h1, x1 = _solve(cga, A, bh, n, 1, fixed_iters=9000)
    assert rep0.iterations == rep1.iterations == 9000
    assert same_bits_nan(h1, h0) and same_bits_nan(x1, x0)
    assert np.isfinite(h1[:4200, 0]).all()        # well past the first slide of the window


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("dict_on", [1, 0])
def test_every_width_with_and_without_coded_columns(cga, monkeypatch, K, dict_on):
    """Every compile-time width of the resident launch, with dictionary-coded and with int32 columns: banded matrices with
    K off-diagonals at mixed distances (inside the tile, in the next tile, many tiles away), symmetric pattern where K is even.
    30 iterations, both paths, same bits (the two paths share no reduction code: any difference in order would show)."""
    from gpu_common import setknob
    setknob(monkeypatch, "CGH_ELL_DICT", dict_on)
    n = 9000                                     # 18 tiles, ragged last one
    offs = [1, -1, 40, -40, 700, -700, 4100, -4100][:K]
    rng = np.random.default_rng(100 + K)
    cols = np.full((K, n), -1, dtype=np.int32)
    vals = np.zeros((K, n))
    i = np.arange(n)
    for s, o in enumerate(offs):
        ok = (i + o >= 0) & (i + o < n)
        cols[s, ok] = (i + o)[ok]
        vals[s, ok] = -rng.random(ok.sum()) * 0.5
    diag = K + 1.0 + rng.random(n)
    bh = rng.standard_normal(n)
    A = cga.Matrix.from_ell(n, K, diag, vals.reshape(-1), cols.reshape(-1))
    rep0, h0, x0 = _solve(cga, A, bh, n, 0, fixed_iters=30)
    rep1, h1, x1 = _solve(cga, A, bh, n, 1, fixed_iters=30)
    assert (A.dict_offsets > 0) == bool(dict_on)
    assert same_bits_nan(h1, h0) and same_bits_nan(x1, x0)
    assert np.isfinite(h1[1:5]).all() and np.isfinite(h1[:5, :2]).all()
