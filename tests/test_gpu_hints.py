"""The cache-hint policy (cgh_fits_llc: plain loads/stores when an iteration's working set fits the Infinity Cache,
non-temporal ones when it streams) must not change a single bit: the same solves in a child process with the streaming
instantiations forced (CGH_LLC_MB=0) and in one with the plain ones forced (CGH_LLC_MB=100000)."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import conjugate_gradient_amd as cga
cga.init(0)
h = hashlib.sha256()
for shape, pts in (((40, 32, 24), 7), ((20, 18, 16), 27), ((64, 64, 32), 7)):
    n = shape[0] * shape[1] * shape[2]
    for strict in (0, 1):
        b = cga.DeviceVector(n); x = cga.DeviceVector(n)
        A = cga.Matrix.poisson(*shape, pts, 0x5EED0001, cga.LAYOUT_ELL, b)
        rep, hist = cga.cg_solve(A, b, x, fixed_iters=25, strict=strict)
        h.update(np.nan_to_num(hist).tobytes()); h.update(x.to_host().tobytes())
print("DIGEST", h.hexdigest())
""" % ROOT


def _digest(env):
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("DIGEST")]
    assert line, p.stdout[-2000:]
    return line[0]


def test_hints_do_not_change_bits():
    streaming = _digest({"CGH_LLC_MB": "0"})
    plain = _digest({"CGH_LLC_MB": "100000"})
    default = _digest({})
    assert streaming == plain == default
