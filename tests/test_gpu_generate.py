"""GPU parity: device generators vs their numpy twins pushed through the ORACLE's converter
(bit-exact), plus size-independent properties at the BASELINE size (256^3)."""
import numpy as np
import pytest

import foamio
import oracle_lib as O
from gpu_common import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    return m


@pytest.mark.parametrize("shape", [(16, 16, 16, 7), (32, 24, 20, 7), (12, 10, 8, 27), (1, 1, 5, 7), (2, 1, 3, 27),
                                   (5, 1, 1, 27), (2, 2, 2, 27), (3, 2, 1, 7)])
def test_poisson_ell_matches_reference_conversion(cga, shape):
    nx, ny, nz, pts = shape
    seed = 0x5EED0001
    s = foamio.poisson_coo(nx, ny, nz, seed, pts)
    n = s["n"]
    b = cga.DeviceVector(n)
    A = cga.Matrix.poisson(nx, ny, nz, pts, seed, cga.LAYOUT_ELL, b)
    K = O.max_nnz_full(n, s["rows"], s["cols"]) if s["U"] else 0
    assert (A.n, A.K, A.nnz_offdiag) == (n, K, 2 * s["U"])
    if K:
        ev, ec, st = O.coo_to_ell_full(n, s["upper"], s["rows"], s["cols"], K)
        gv, gc = A.download_ell()
        assert st == 0 and np.array_equal(gc, ec) and np.array_equal(bits(gv), bits(ev))
    assert np.array_equal(bits(A.download_diag()), bits(s["diag"]))
    assert np.array_equal(bits(b.to_host()), bits(s["source"]))
