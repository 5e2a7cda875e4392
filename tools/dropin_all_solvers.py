"""All five solvers of include/cg.h through host pointers on the largest OpenFOAM system of the reference (cylinder_32k):
libcg_hip.so against the reference on one host core, same return value and x bits expected (CG_HIP_QUIET=1)."""
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
G = cga.lib()
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref.so"))
tag = sys.argv[1] if len(sys.argv) > 1 else "32k"
s = foamio.foam(tag)
n, U = s["n"], s["U"]
diag, upper, rows, cols, b = s["diag"], s["upper"], s["rows"], s["cols"], s["source"]
Kf = O.max_nnz_full(n, rows, cols)
evf, ecf, _ = O.coo_to_ell_full(n, upper, rows, cols, Kf)
row_ptr = np.zeros(n + 1, dtype=np.int32)
np.add.at(row_ptr, rows + 1, 1)
row_ptr = np.cumsum(row_ptr).astype(np.int32)
Ku = int(np.bincount(rows, minlength=n).max())
evu = np.zeros(n * Ku); ecu = np.full(n * Ku, -1, dtype=np.int32); evr = np.zeros(n * Ku); ecr = np.full(n * Ku, -1, dtype=np.int32)
fill = np.zeros(n, dtype=np.int64)
for k in range(U):
    i = rows[k]; sl = fill[i]; fill[i] += 1
    evu[sl * n + i] = upper[k]; ecu[sl * n + i] = cols[k]      # slot-major upper ELL
    evr[i * Ku + sl] = upper[k]; ecr[i * Ku + sl] = cols[k]    # row-major upper ELL
calls = {
    "conjugate_gradient_coo": lambda L, x, mi: L.conjugate_gradient_coo(n, D(diag), U, D(upper), I(rows), I(cols), D(b), D(x), mi, C.c_double(1e-6)),
    "conjugate_gradient_csr": lambda L, x, mi: L.conjugate_gradient_csr(n, D(diag), U, D(upper), I(row_ptr), I(cols), D(b), D(x), mi, C.c_double(1e-6)),
    "conjugate_gradient_ell": lambda L, x, mi: L.conjugate_gradient_ell(n, D(diag), U, D(evr), I(ecr), Ku, D(b), D(x), mi, C.c_double(1e-6)),
    "conjugate_gradient_ell_colmajor": lambda L, x, mi: L.conjugate_gradient_ell_colmajor(n, D(diag), U, D(evu), I(ecu), Ku, D(b), D(x), mi, C.c_double(1e-6)),
    "conjugate_gradient_ell_full_colmajor": lambda L, x, mi: L.conjugate_gradient_ell_full_colmajor(n, D(diag), U, D(evf), I(ecf), Kf, D(b), D(x), mi, C.c_double(1e-6)),
}
for name, f in calls.items():
    xg = np.zeros(n); xr = np.zeros(n)
    f(G, xg, 3)
    t0 = time.perf_counter(); rg = f(G, xg, 1000); tg = time.perf_counter() - t0
    with quiet_stdout():
        t0 = time.perf_counter(); rr = f(R, xr, 1000); tr = time.perf_counter() - t0
    print("%-38s libcg_hip %8.1f ms   reference %8.1f ms   returns %d / %d   same x bits: %s" % (name, tg * 1e3, tr * 1e3, rg, rr, xg.tobytes() == xr.tobytes()), flush=True)
