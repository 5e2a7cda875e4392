"""GPU parity of CGH_LAYOUT_PB, the two-phase (propagation-blocking) SpMV added for BASELINE config 5 (spmv_pb.hip).

A throughput layout: phase 1 multiplies every stored value by its x (staged in LDS) and streams the products, phase 2
adds them into row sums held in LDS with LDS atomics -- in the order of the product stream (the default since the end of
round 3: the same bits on every run) or, ordered = 0, as the atomics retire; neither is the reference's order.  The reference's
SpMV for such a matrix is mv_csr_symmetric / mv_coo_symmetric (src/csr.c:190-229, src/coo.c:202-233); the bar here is
the throughput-mode bar of the north star: every y_i within a few ulps of sum_j |a_ij x_j| of the reference's value
(goldens produced by oracle/_ref), CG iterations within +-1 and the residual trajectory within 1e-10 relative over the
first iterations; strict=1 is refused (bit-identical callers use CSR / ELL / sliced ELL)."""
import numpy as np
import pytest

import foamio
import oracle_lib as O
from gpu_common import GOLDENS, golden, inputs

pytestmark = pytest.mark.gpu

PB_CONFIGS = [  # log2 of the rows per block, entries per block aimed at, workgroups per column block in phase 1
    (14, 131072, 1),    # the defaults
    (13, 300, 1),       # every 256-row group a block of its own: many tiny segments, empty segments
    (14, 3000, 3),      # three workgroups share a column block
]


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    yield m
    m.check(m.lib().cgh_config_reset())


def _abs_bound(n, diag, vals, rows, cols, x):
    """sum_j |a_ij x_j| per row (the scale of a row sum's rounding error)"""
    s = np.abs(diag * x)
    np.add.at(s, rows, np.abs(vals * x[cols]))
    np.add.at(s, cols, np.abs(vals * x[rows]))
    return s


@pytest.mark.parametrize("ordered", [1, 0])
@pytest.mark.parametrize("cfg", PB_CONFIGS)
@pytest.mark.parametrize("tag", GOLDENS)
def test_pb_spmv_matches_reference_to_rounding(cga, tag, cfg, ordered):
    s, g = inputs(tag), golden(tag)
    n = s["n"]
    cga.config_pb(*cfg, ordered)
    A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_PB)
    assert (A.n, A.K, A.nnz_offdiag, A.layout) == (n, int(g["K_full"]), 2 * s["U"], cga.LAYOUT_PB)
    x = cga.DeviceVector.from_host(g["x_probe"]); y = cga.DeviceVector(n)
    bound = _abs_bound(n, s["diag"], s["upper"], s["rows"], s["cols"], g["x_probe"])
    for _ in range(3):                                   # the order of additions may differ from run to run: same bar every time
        A.spmv(x, y)
        assert np.all(np.abs(y.to_host() - g["y_probe"]) <= 8 * np.finfo(float).eps * bound + 1e-300)
    with pytest.raises(RuntimeError):
        A.download_ell()                                 # the layout does not keep the rows' entry order


@pytest.mark.parametrize("cfg", PB_CONFIGS)
def test_pb_skewed_rows_and_shapes(cga, cfg):
    """power-law rows up to 300 entries, n not a multiple of the 256-row partition unit, then the edge shapes: fewer
    rows than one unit, a diagonal matrix (no entries at all), one dense row"""
    cga.config_pb(*cfg, 1)
    n, max_len, seed = 6001, 300, 0x5EED0005
    s = foamio.powerlaw_coo(n, max_len, seed)
    rng = np.random.default_rng(3)
    diag = 1.0 + rng.random(n) * 40; xh = rng.standard_normal(n)
    want = O.mv_coo(n, diag, s["upper"], s["rows"], s["cols"], xh)
    A = cga.Matrix.from_coo(n, diag, s["upper"], s["rows"], s["cols"], cga.LAYOUT_PB)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    A.spmv(x, y)
    bound = _abs_bound(n, diag, s["upper"], s["rows"], s["cols"], xh)
    assert np.all(np.abs(y.to_host() - want) <= 16 * np.finfo(float).eps * bound)
    for n2, rows, cols in ((5, [0, 0, 1, 3], [1, 4, 2, 4]), (70, [], []), (300, [0] * 299, list(range(1, 300))),
                           (1000, [0] * 999 + [255, 256, 511], list(range(1, 1000)) + [256, 999, 512])):
        rows = np.array(rows, dtype=np.int32); cols = np.array(cols, dtype=np.int32)
        vals = -(1.0 + np.arange(rows.size)) / 7.0
        d2 = 3.0 + np.arange(n2) * 0.5; x2 = 1.0 / (1.0 + np.arange(n2))
        want2 = O.mv_coo(n2, d2, vals, rows, cols, x2) if rows.size else d2 * x2
        A2 = cga.Matrix.from_coo(n2, d2, vals, rows, cols, cga.LAYOUT_PB)
        xd = cga.DeviceVector.from_host(x2); yd = cga.DeviceVector(n2)
        A2.spmv(xd, yd)
        assert np.allclose(yd.to_host(), want2, rtol=1e-14, atol=0), (n2, cfg)


@pytest.mark.parametrize("ordered", [1, 0])
def test_pb_pads_do_not_leak_non_finite_x(cga, ordered):
    """segments are padded with (value 0, column 0 of the block) entries whose products are never added: an infinite or
    NaN x[0] must reach only the rows that really have a column-0 entry (and row 0 through its diagonal)"""
    cga.config_pb(14, 300, 1, ordered)   # (ordered: the pads' products go to spare LDS slots; not: they are skipped)
    n = 1500
    s = foamio.powerlaw_coo(n, 40, 0x5EED0005)
    diag = np.full(n, 50.0); xh = np.ones(n); xh[0] = np.inf; xh[256] = np.nan
    A = cga.Matrix.from_coo(n, diag, s["upper"], s["rows"], s["cols"], cga.LAYOUT_PB)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    A.spmv(x, y)
    got = y.to_host()
    touched = np.zeros(n, dtype=bool); touched[[0, 256]] = True
    for bad in (0, 256):
        touched[s["cols"][s["rows"] == bad]] = True
        touched[s["rows"][s["cols"] == bad]] = True
    assert np.all(np.isfinite(got[~touched])) and not np.any(np.isfinite(got[touched]))


@pytest.mark.parametrize("cfg", PB_CONFIGS)
def test_pb_ordered_mode_gives_the_same_bits_every_run(cga, cfg):
    """cgh_config_pb(..., ordered=1): phase 2 adds the product stream's pieces in stream order -- SpMV output and a whole
    CG trajectory bit-identical from run to run (the order in which the LDS atomics retire, ordered=0, is not), same accuracy.
    The default since the end of round 3."""
    cga.config_pb(*cfg, 1)
    n, max_len, seed = 40001, 500, 0x5EED0005
    s = foamio.powerlaw_coo(n, max_len, seed)
    rng = np.random.default_rng(4)
    diag = 1.0 + np.bincount(s["rows"], np.abs(s["upper"]), n) + np.bincount(s["cols"], np.abs(s["upper"]), n)
    xh = rng.standard_normal(n)
    A = cga.Matrix.from_coo(n, diag, s["upper"], s["rows"], s["cols"], cga.LAYOUT_PB)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    A.spmv(x, y)
    first = y.to_host()
    want = O.mv_coo(n, diag, s["upper"], s["rows"], s["cols"], xh)
    assert np.all(np.abs(first - want) <= 16 * np.finfo(float).eps * _abs_bound(n, diag, s["upper"], s["rows"], s["cols"], xh))
    for _ in range(6):
        A.spmv(x, y)
        assert np.array_equal(y.to_host().view(np.uint64), first.view(np.uint64))
    b = cga.DeviceVector.from_host(rng.standard_normal(n)); xs = cga.DeviceVector(n)
    rep, hist = cga.cg_solve(A, b, xs, max_iter=60, tol=1e-9, strict=0)
    rep2, hist2 = cga.cg_solve(A, b, xs, max_iter=60, tol=1e-9, strict=0)
    assert rep2.ret == rep.ret and hist.tobytes() == hist2.tobytes()


@pytest.mark.parametrize("tag", ["cylinder_2000", "poisson7_32x24x20", "poisson27_12x10x8"])
def test_pb_cg_throughput_mode_and_strict_refused(cga, tag):
    s, g = inputs(tag), golden(tag)
    n = s["n"]
    cga.config_pb(13, 2000, 2)
    A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_PB)
    b = cga.DeviceVector.from_host(s["source"]); x = cga.DeviceVector(n)
    rep, hist = cga.cg_solve(A, b, x, max_iter=int(g["max_iter"]), tol=float(g["tol"]), strict=0)
    assert abs(rep.ret - int(g["cg_ret"][4])) <= 1
    k = min(11, hist.shape[0], g["traj"].shape[0])
    rel = np.abs(hist[:k, 0] - g["traj"][:k, 0]) / np.abs(g["traj"][:k, 0])
    assert rel.max() < 1e-10, rel.max()                  # north star: 1e-10 relative (first iterations, throughput mode)
    with pytest.raises(RuntimeError):
        cga.cg_solve(A, b, x, max_iter=10, tol=1e-6, strict=1)
