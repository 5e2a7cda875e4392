"""Fixed cost of one call through the reference's host-pointer API (conjugate_gradient_ell_full_colmajor on the
OpenFOAM cylinder_32k system): wall time for max_iter = 2, 34, 100, 1000.  CG_HIP_MODE=fast|strict selects the reductions."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CG_HIP_QUIET"] = "1"
import conjugate_gradient_amd as cga, foamio, oracle_lib as O
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
D = lambda a: a.ctypes.data_as(dp); I = lambda a: a.ctypes.data_as(ip)
G = cga.lib()
G.conjugate_gradient_ell_full_colmajor.argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
s = foamio.foam("32k"); n = s["n"]
K = O.max_nnz_full(n, s["rows"], s["cols"])
ev, ec, st = O.coo_to_ell_full(n, s["upper"], s["rows"], s["cols"], K)
x = np.zeros(n)
for mi in (2, 2, 34, 100, 1000):
    t0 = time.perf_counter()
    ret = G.conjugate_gradient_ell_full_colmajor(n, D(s["diag"]), s["U"], D(ev), I(ec), K, D(s["source"]), D(x), mi, 1e-6)
    print("max_iter %5d: %8.2f ms (ret %d)" % (mi, (time.perf_counter() - t0) * 1e3, ret), flush=True)
