"""Time to solution through the REFERENCE'S OWN API (host pointers in, host pointers out): the reference's
conjugate_gradient_ell_full_colmajor on one host core (oracle/_ref/libref.so) against the same symbol of libcg_hip.so
(upload + dictionary build + bit-identical solve + download; CG_HIP_QUIET=1 so neither side pays for stdout).
Cases: the largest OpenFOAM system the reference ships (cylinder_32k, converges in max_iter) and the 256^3 Poisson matrix
for a fixed 60 iterations (tol < 0).  PCIe-inclusive by construction."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CG_HIP_QUIET"] = "1"
import conjugate_gradient_amd as cga  # noqa: E402
import foamio  # noqa: E402
import oracle_lib as O  # noqa: E402
from bench import quiet_stdout  # noqa: E402

dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
D = lambda a: a.ctypes.data_as(dp)  # noqa: E731
I = lambda a: a.ctypes.data_as(ip)  # noqa: E731
sig = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
G = cga.lib()
G.conjugate_gradient_ell_full_colmajor.argtypes = sig
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref.so"))
R.conjugate_gradient_ell_full_colmajor.argtypes = sig


def case(name, n, diag, ev, ec, K, U, b, max_iter, tol, cpu_iters=None):
    x_g = np.zeros(n); x_r = np.zeros(n)
    G.conjugate_gradient_ell_full_colmajor(n, D(diag), U, D(ev), I(ec), K, D(b), D(x_g), 3, tol)        # warm-up (context, code load)
    t0 = time.perf_counter()
    ret_g = G.conjugate_gradient_ell_full_colmajor(n, D(diag), U, D(ev), I(ec), K, D(b), D(x_g), max_iter, tol)
    t_g = time.perf_counter() - t0
    mi = cpu_iters if cpu_iters else max_iter
    with quiet_stdout():
        t0 = time.perf_counter()
        ret_r = R.conjugate_gradient_ell_full_colmajor(n, D(diag), U, D(ev), I(ec), K, D(b), D(x_r), mi, tol)
        t_r = time.perf_counter() - t0
    its_g = (ret_g - 1) if ret_g > 0 else max_iter - 1
    its_r = (ret_r - 1) if ret_r > 0 else mi - 1
    same = (ret_g == ret_r and x_g.tobytes() == x_r.tobytes()) if mi == max_iter else None
    print("%-28s n=%9d  libcg_hip %9.1f ms (%5d iterations, %7.1f us each incl. transfers)   reference, 1 core %10.1f ms (%5d iterations,"
          " %9.1f us each)   same return value and x bits: %s" % (name, n, t_g * 1e3, its_g, t_g / its_g * 1e6, t_r * 1e3, its_r,
                                                                  t_r / its_r * 1e6, same), flush=True)


s = foamio.foam("32k")
n = s["n"]
K = O.max_nnz_full(n, s["rows"], s["cols"])
ev, ec, st = O.coo_to_ell_full(n, s["upper"], s["rows"], s["cols"], K)
case("OpenFOAM cylinder_32k", n, s["diag"], ev, ec, K, s["U"], s["source"], 1000, 1e-6)

cga.init(0)
N = 256
b = cga.DeviceVector(N ** 3)
A = cga.Matrix.poisson(N, N, N, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
ev, ec = A.download_ell(); diag = A.download_diag(); bh = b.to_host(); U = int(A.nnz_offdiag // 2); K = A.K
A.free(); b.free()
case("7-point Poisson 256^3", N ** 3, diag, ev, ec, K, U, bh, 61, -1.0, cpu_iters=11)
