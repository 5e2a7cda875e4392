"""Device-memory leak check of the host-pointer entry points: free HBM before and after 450 calls (hipMemGetInfo)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CG_HIP_QUIET"] = "1"
import conjugate_gradient_amd as cga, foamio, oracle_lib as O
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
D = lambda a: a.ctypes.data_as(dp); I = lambda a: a.ctypes.data_as(ip)
G = cga.lib()
G.conjugate_gradient_ell_full_colmajor.argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
G.conjugate_gradient_csr.argtypes = [C.c_int, dp, C.c_int, dp, ip, ip, dp, dp, C.c_int, C.c_double]
G.mv_ell_symmetric_full_colmajor.argtypes = [C.c_int, C.c_int, dp, dp, ip, dp, dp]
s = foamio.foam("8000"); n = s["n"]
K = O.max_nnz_full(n, s["rows"], s["cols"])
ev, ec, st = O.coo_to_ell_full(n, s["upper"], s["rows"], s["cols"], K)
row_ptr = np.zeros(n + 1, dtype=np.int32); np.add.at(row_ptr, s["rows"] + 1, 1); row_ptr = np.cumsum(row_ptr).astype(np.int32)
x = np.zeros(n); y = np.zeros(n)
G.conjugate_gradient_ell_full_colmajor(n, D(s["diag"]), s["U"], D(ev), I(ec), K, D(s["source"]), D(x), 50, 1e-6)
f0 = free_bytes()
done = 0
for target in (150, 600, 1500):
    for k in range(target - done):
        G.conjugate_gradient_ell_full_colmajor(n, D(s["diag"]), s["U"], D(ev), I(ec), K, D(s["source"]), D(x), 40, 1e-6)
        G.conjugate_gradient_csr(n, D(s["diag"]), s["U"], D(s["upper"]), I(row_ptr), I(s["cols"]), D(s["source"]), D(x), 40, 1e-6)
        G.mv_ell_symmetric_full_colmajor(n, K, D(s["diag"]), D(ev), I(ec), D(s["source"]), D(y))
    done = target
    print("after %5d rounds of 3 calls: free HBM down by %d bytes" % (done, f0 - free_bytes()), flush=True)
