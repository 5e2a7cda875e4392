"""GPU parity of the two layouts added for BASELINE config 5 (skewed rows, scattered columns):

* CGH_LAYOUT_SELLS, the sigma-sorted sliced ELL: rows sorted by length inside windows before slicing, the permutation
  folded into y.  Per-row order is the reference's, so SpMV and the strict CG trajectory are bit-identical to the
  goldens (reference src/ell.c:1004-1040, src/cg.c:414-508 through oracle/_ref).
* CGH_LAYOUT_CSRB, the column-blocked CSR: a throughput layout -- a row's products are added slab by slab, a fixed
  order that is not the reference's.  Bar: every y_i within a few ulps of sum|a_ij x_j| of the reference's value,
  bit-identical run to run, CG iterations within +-1 and the residual trajectory within 1e-10 relative over the first
  iterations (north star); strict=1 is refused."""
import numpy as np
import pytest

import foamio
import oracle_lib as O
from gpu_common import GOLDENS, bits, golden, inputs, same_bits_nan

pytestmark = pytest.mark.gpu

CSRB_CONFIGS = [  # rows per wave-block, slab bits, workgroups per CU, window, quorum divisor, LDS atomics
    (1024, 17, 0, 3, 2, 0),     # the defaults
    (512, 6, 0, 3, 2, 0),       # many tiny slabs: chunks span several slabs, empty slabs, every gate exercised
    (2048, 9, 0, 2, 16, 0),     # strict quorum
    (1024, 8, 1, 0, 2, 0),      # no pacing, one workgroup per CU (more rounds)
    (1024, 7, 0, 3, 2, 1),      # LDS atomics instead of scan + read-add-write
]


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    yield m
    m.config_csrb(1024, 17, 0, 3, 2, 0)
    m.config_sell_sigma(65536)


def _abs_bound(n, diag, vals, rows, cols, x):
    """sum_j |a_ij x_j| per row (the scale of a row sum's rounding error)"""
    s = np.abs(diag * x)
    np.add.at(s, rows, np.abs(vals * x[cols]))
    np.add.at(s, cols, np.abs(vals * x[rows]))
    return s


@pytest.mark.parametrize("cfg", CSRB_CONFIGS)
@pytest.mark.parametrize("tag", GOLDENS)
def test_csrb_spmv_matches_reference_to_rounding(cga, tag, cfg):
    s, g = inputs(tag), golden(tag)
    n = s["n"]
    cga.config_csrb(*cfg)
    A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_CSRB)
    assert (A.n, A.K, A.nnz_offdiag, A.layout) == (n, int(g["K_full"]), 2 * s["U"], cga.LAYOUT_CSRB)
    x = cga.DeviceVector.from_host(g["x_probe"]); y = cga.DeviceVector(n)
    A.spmv(x, y)
    got = y.to_host()
    bound = _abs_bound(n, s["diag"], s["upper"], s["rows"], s["cols"], g["x_probe"])
    assert np.all(np.abs(got - g["y_probe"]) <= 8 * np.finfo(float).eps * bound + 1e-300)
    for _ in range(2):                                   # the order of additions is fixed: same bits every run
        A.spmv(x, y)
        assert np.array_equal(bits(y.to_host()), bits(got))
    with pytest.raises(RuntimeError):
        A.download_ell()                                 # the layout does not keep the rows' entry order


@pytest.mark.parametrize("cfg", CSRB_CONFIGS[:3])
def test_csrb_skewed_rows_and_shapes(cga, cfg):
    """power-law rows up to 300 entries (runs of one row inside a slab, rows spanning all slabs), n not a multiple of 64,
    then the edge shapes: fewer rows than one segment, a diagonal matrix, one dense row"""
    cga.config_csrb(*cfg)
    n, max_len, seed = 6001, 300, 0x5EED0005
    s = foamio.powerlaw_coo(n, max_len, seed)
    rng = np.random.default_rng(3)
    diag = 1.0 + rng.random(n) * 40; xh = rng.standard_normal(n)
    want = O.mv_coo(n, diag, s["upper"], s["rows"], s["cols"], xh)
    A = cga.Matrix.from_coo(n, diag, s["upper"], s["rows"], s["cols"], cga.LAYOUT_CSRB)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    A.spmv(x, y)
    bound = _abs_bound(n, diag, s["upper"], s["rows"], s["cols"], xh)
    assert np.all(np.abs(y.to_host() - want) <= 16 * np.finfo(float).eps * bound)
    for n2, rows, cols in ((5, [0, 0, 1, 3], [1, 4, 2, 4]), (70, [], []), (300, [0] * 299, list(range(1, 300)))):
        rows = np.array(rows, dtype=np.int32); cols = np.array(cols, dtype=np.int32)
        vals = -(1.0 + np.arange(rows.size)) / 7.0
        d2 = 3.0 + np.arange(n2) * 0.5; x2 = 1.0 / (1.0 + np.arange(n2))
        want2 = O.mv_coo(n2, d2, vals, rows, cols, x2) if rows.size else d2 * x2
        A2 = cga.Matrix.from_coo(n2, d2, vals, rows, cols, cga.LAYOUT_CSRB)
        xd = cga.DeviceVector.from_host(x2); yd = cga.DeviceVector(n2)
        A2.spmv(xd, yd)
        assert np.allclose(yd.to_host(), want2, rtol=1e-14, atol=0), (n2, cfg)


@pytest.mark.parametrize("tag", ["cylinder_2000", "poisson7_32x24x20", "poisson27_12x10x8"])
def test_csrb_cg_throughput_mode_and_strict_refused(cga, tag):
    s, g = inputs(tag), golden(tag)
    n = s["n"]
    cga.config_csrb(512, 7, 0, 3, 2, 0)
    A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_CSRB)
    b = cga.DeviceVector.from_host(s["source"]); x = cga.DeviceVector(n)
    rep, hist = cga.cg_solve(A, b, x, max_iter=int(g["max_iter"]), tol=float(g["tol"]), strict=0)
    assert abs(rep.ret - int(g["cg_ret"][4])) <= 1
    k = min(11, hist.shape[0], g["traj"].shape[0])
    rel = np.abs(hist[:k, 0] - g["traj"][:k, 0]) / np.abs(g["traj"][:k, 0])
    assert rel.max() < 1e-10, rel.max()                  # north star: 1e-10 relative (first iterations, throughput mode)
    rep2, hist2 = cga.cg_solve(A, b, x, max_iter=int(g["max_iter"]), tol=float(g["tol"]), strict=0)
    assert rep2.ret == rep.ret and same_bits_nan(hist, hist2)   # deterministic trajectory
    with pytest.raises(RuntimeError):
        cga.cg_solve(A, b, x, max_iter=10, tol=1e-6, strict=1)


@pytest.mark.parametrize("sigma", [64, 256, 65536])
@pytest.mark.parametrize("tag", GOLDENS)
def test_sells_conversion_spmv_and_strict_cg_bit_exact(cga, tag, sigma):
    s, g = inputs(tag), golden(tag)
    n = s["n"]
    cga.config_sell_sigma(sigma)
    A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_SELLS)
    assert (A.n, A.K, A.nnz_offdiag, A.layout) == (n, int(g["K_full"]), 2 * s["U"], cga.LAYOUT_SELLS)
    ev, ec = A.download_ell()                            # un-permuted: equals the reference's conversion
    assert np.array_equal(ec, g["ell_full_cols"]) and np.array_equal(bits(ev), bits(g["ell_full_vals"]))
    x = cga.DeviceVector.from_host(g["x_probe"]); y = cga.DeviceVector(n)
    A.spmv(x, y)
    assert np.array_equal(bits(y.to_host()), bits(g["y_probe"]))
    if sigma == 256 and tag in ("cylinder_2000", "poisson7_16", "poisson27_12x10x8"):
        b = cga.DeviceVector.from_host(s["source"]); xs = cga.DeviceVector(n)
        rep, hist = cga.cg_solve(A, b, xs, max_iter=int(g["max_iter"]), tol=float(g["tol"]), strict=1)
        assert rep.ret == int(g["cg_ret"][4]) and same_bits_nan(hist, g["traj"])
        assert np.array_equal(bits(xs.to_host()), bits(g["cg_x"]))


def test_sells_padding_shrinks_with_sigma(cga):
    """power-law rows: unsorted slices of 64 rows pad to the longest row of each slice; sorting windows remove most of it"""
    n, max_len, seed = 60000, 1000, 0x5EED0005
    b = cga.DeviceVector(n)
    csr = cga.Matrix.powerlaw(n, max_len, seed, cga.LAYOUT_CSR, b)
    rng = np.random.default_rng(8)
    xh = rng.standard_normal(n)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    csr.set_exact_rows(True)
    csr.spmv(x, y); want = y.to_host()
    plain = cga.Matrix.powerlaw(n, max_len, seed, cga.LAYOUT_SELL, b)
    ratios = []
    for sigma in (256, 4096, 65536):
        cga.config_sell_sigma(sigma)
        A = cga.Matrix.powerlaw(n, max_len, seed, cga.LAYOUT_SELLS, b)
        A.spmv(x, y)
        assert np.array_equal(bits(y.to_host()), bits(want))
        ratios.append(A.stored_bytes / csr.stored_bytes)
        A.free()
    assert plain.stored_bytes / csr.stored_bytes > 5.0
    assert ratios[0] > ratios[1] > ratios[2] and ratios[2] < 1.5, ratios
